#!/usr/bin/env python3
"""Achieved HBM bandwidth of the element-wise / pooling / loss / optimizer kernels of one training step (BASELINE configs[1]:
batch 64, 128x64, base 32, depth 3, bf16) from a rocprofv3 kernel trace of bench.py.
usage: python tools/elementwise_roofline.py results.db > profiles/xxx.txt
Algorithmic bytes = every tensor the kernel must read once + write once, T(level) = B*H*W*C*2 bytes."""
import re, sqlite3, sys

B, PEAK = 64, 8000.0
LEVELS = [0, 1, 2, 3, 2, 1, 0]                                                # block order, forward and backward
NPARAM = 2_199_720

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "adamw" in r[0]]
nsteps = min(10, len(idx) - 1)
per = idx[-1] - idx[-2]
# element size of the activations: the fp32 tier's kernels are the <float, ...> instantiations
ESZ = 4 if any("_kernel<float" in r[0] or "IfLi" in r[0] or "IfEE" in r[0] for r in rows[idx[-2]:idx[-1]]) else 2
T = [B * (128 >> l) * (64 >> l) * (32 << l) * ESZ for l in range(4)]        # bytes of one feature map per level

def avg(pos):
    d = [rows[idx[-k - 1] + 1 + pos][2] - rows[idx[-k - 1] + 1 + pos][1] for k in range(1, nsteps + 1)]
    return sum(d) / len(d) / 1e3

def kname(n):
    m = re.search(r"(?:sx::|_ZN2sx\d+)([A-Za-z0-9_]+?)_kernel", n)
    return m.group(1) if m else n[:30]

step = [(pos, rows[idx[-2] + 1 + pos][0]) for pos in range(per)]
out = []
fi = 0               # forward block counter (two launches per block)
bblk = bstate = 0    # backward block counter and position inside the block
fwd_in_block = 0
pool_f = pool_b = 0
for pos, name in step:
    k = kname(name)
    us = avg(pos)
    byt, what = None, ""
    if k in ("act_fwd", "act_fwd_pool"):
        lvl = LEVELS[fi]
        if fwd_in_block == 0:
            byt, what = 2 * T[lvl], f"GN+SiLU fwd            L{lvl}: read Y, write A"
            fwd_in_block = 1
        else:
            if k == "act_fwd_pool":
                byt, what = (3.25 + 0.25 / ESZ) * T[lvl], f"block out + max-pool   L{lvl}: read Y,S, write O,O/4,positions"
            else:
                byt, what = 3 * T[lvl], f"block out fwd          L{lvl}: read Y,S, write O"
            fwd_in_block = 0
            fi += 1
    elif k in ("act_bwd_reduce", "act_bwd_apply"):
        # per DoubleConv block, backward: block-output sums, block-output apply, [GroupNorm-1 sums unless conv2's data gradient
        # took them in its epilogue], GroupNorm-1 apply
        lvl = LEVELS[bblk]
        if bstate == 0: byt, what, bstate = 3 * T[lvl], f"block out bwd, sums    L{lvl}: read Y,dO,S", 1
        elif bstate == 1: byt, what, bstate = 4 * T[lvl], f"block out bwd, apply   L{lvl}: read Y,dO, write dY,dR", 2
        elif bstate == 2 and k == "act_bwd_reduce": byt, what, bstate = 2 * T[lvl], f"GN+SiLU bwd, sums      L{lvl}: read Y,dO", 3
        else:
            byt, what = 3 * T[lvl], f"GN+SiLU bwd, apply     L{lvl}: read Y,dO, write dY"
            bstate = 0
            bblk += 1
    elif k == "act_bwd_group":
        # both phases of a backward in one launch (small levels): block output (state 0) or GroupNorm-1 (state 2)
        lvl = LEVELS[bblk]
        if bstate == 0: byt, what, bstate = 5 * T[lvl], f"block out bwd, 1 launch L{lvl}: read Y,dO,S, write dY,dR", 2
        else:
            byt, what = 3 * T[lvl], f"GN+SiLU bwd, 1 launch  L{lvl}: read Y,dO, write dY"
            bstate = 0
            bblk += 1
    elif k in ("maxpool_bwd_add", "maxpool_bwd_add_idx"):
        lvl = [2, 1, 0][pool_b]; pool_b += 1
        if k.endswith("idx"): byt, what = (2.25 + 0.25 / ESZ) * T[lvl], f"max-pool bwd + skip    L{lvl}: read positions,dskip,dP, write dO"
        else: byt, what = 3.25 * T[lvl], f"max-pool bwd + skip    L{lvl}: read O,dskip,dP, write dO"
    elif k == "maxpool_fwd":
        lvl = [0, 1, 2][pool_f]; pool_f += 1
        byt, what = 1.25 * T[lvl], f"max-pool fwd           L{lvl}"
    elif k == "adamw":
        byt, what = 28 * NPARAM, "AdamW (p,g,m,v read; p,m,v written)"
    elif k == "sumsq":
        byt, what = 4 * NPARAM, "gradient norm (read g)"
    elif k == "base_pool":
        byt, what = B * 128 * 64 * (8 * 2 + 8 * 4 + 4 + 8 * 4) + 4 * B * 64 * 32 * 8 * 4, "loss: base+edge terms, planar gradient, pooled fields"
    elif k == "grad_out":
        byt, what = B * 128 * 64 * (8 * 4 + 8 * 2) + 2 * B * 64 * 32 * 8 * 4, "loss: transposed Sobel + NHWC gradient write"
    elif k == "loss_tile":
        byt, what = B * 128 * 64 * (8 * ESZ + 8 * 4 + 4 + 8 * ESZ) + 4 * B * 64 * 32, "loss, one tile kernel: read pred, target, mask, pooled mask; write d(pred)"
    elif k == "sobel_fwd":
        byt, what = 6 * B * 64 * 32 * 8 * 4, "loss: Sobel of pooled pred/target + its gradient"
    elif k == "wgrad_reduce_multi":
        what = "slab reduction (bytes depend on the split plan)"
    if byt is not None:
        gbs = byt / us / 1e3
        out.append(f"{pos:4d} {k:18s} {us:7.2f} us  {byt/1e6:8.1f} MB  {gbs:7.0f} GB/s  {100*gbs/PEAK:5.1f} %   {what}")
print(f"# element-wise / pooling / loss / optimizer kernels of one training step ({'fp32' if ESZ == 4 else 'bf16'} activations), avg over the last {nsteps} steps of the trace")
print(f"# achieved = algorithmic bytes / kernel time; peak 8000 GB/s (practical streaming rate on this part: 4.5-5 TB/s;")
print(f"# figures above that are re-reads served by the 256 MB MALL)")
print("# pos kernel              time       bytes      achieved   of peak  what")
print("\n".join(out))
tot_b = sum(float(l.split()[4]) for l in out); tot_t = sum(float(l.split()[2]) for l in out)
print(f"# total {tot_t:.0f} us, {tot_b:.0f} MB -> {tot_b/tot_t*1e3:.0f} GB/s aggregate")
