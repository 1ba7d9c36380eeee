"""Device-resident data path: ResidentLoader batches == the reference-style host DataLoader batches
(same order for a torch seed, same values up to the device normalizer's fp32 rounding), and sx_gather_rows alone."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from unet_amd import _lib
    _lib.lib()
    return _lib


def test_gather_rows_matches_index_select(L):
    g = torch.Generator().manual_seed(0)
    for n_src, row, stride_extra, nb in [(17, 64, 0, 9), (5, 7, 3, 5), (40, 8 * 128 * 64, 0, 16), (3, 1, 0, 1)]:
        src = torch.randn(n_src, row + stride_extra, generator=g).cuda()
        idx = torch.randint(0, n_src, (nb,), generator=g)
        dst = torch.full((nb, row + 4), -5.0, device="cuda")
        L.check(L.lib().sx_gather_rows(L.ptr(src), row + stride_extra, n_src, L.ptr(idx.cuda()), nb, row, L.ptr(dst), row + 4,
                                       L.stream_ptr()))
        torch.cuda.synchronize()
        assert torch.equal(dst[:, :row].cpu(), src.cpu()[idx][:, :row])
        assert float((dst[:, row:] + 5.0).abs().max()) == 0.0
    with pytest.raises(ValueError):
        L.check(L.lib().sx_gather_rows(L.ptr(src), 1, 3, L.ptr(idx.cuda()), 70000, 1, L.ptr(dst), 1, L.stream_ptr()))


def _npz(tmp_path, N=37, H=16, W=8):
    rng = np.random.default_rng(3)
    mask = np.zeros((N, H, W), np.float32)
    for i in range(N):
        mask[i, 2 + i % 3:H - 2, 1:W - 1 - i % 2] = 1
    Y = rng.normal(size=(N, 3, H, W)).astype(np.float32) * 10
    Y[:, 0] = np.abs(Y[:, 0])
    Y[1, 0, 3, 3] = np.nan
    Y[2, 2, 4, 2] = np.inf
    path = str(tmp_path / "d.npz")
    np.savez(path, Y=Y, y_keys=np.array(["Te", "ne", "S"]), mask=mask, X=rng.normal(size=(N, 5)).astype(np.float32),
             param_keys=np.array(["Gamma_D2", "Ptot_W", "n_core", "dna", "hci"]))
    return path


@pytest.mark.parametrize("mode", ["params", "autoencoder"])
def test_resident_loader_equals_host_loader(tmp_path, mode):
    from unet_amd import data as D
    path = _npz(tmp_path)

    def loaders(resident):
        norms = dict(Te=D.MaskedLogStandardizer(eps=1.0), S=D.MaskedSymLogStandardizer(scale=5.0))
        return D.make_loaders(path, inputs_mode=mode, batch_size=8, y_keys=["Te", "S"], norms_by_name=norms, device="cuda",
                              resident=resident)
    th, vh, *_ = loaders(False)
    tr, vr, *_ = loaders(True)
    assert isinstance(tr, D.ResidentLoader) and len(tr) == len(th) and len(vr) == len(vh)
    for hl, rl in ((th, tr), (vh, vr)):
        for epoch in range(2):
            torch.manual_seed(100 + epoch)
            hb = list(hl)
            torch.manual_seed(100 + epoch)
            rb = list(rl)
            assert len(hb) == len(rb)
            for a, b in zip(hb, rb):
                assert a["idx"].tolist() == b["idx"].tolist()
                for k in ("x", "y", "mask", "params"):
                    assert b[k].is_cuda and b[k].shape == a[k].shape, k
                    assert torch.allclose(b[k].cpu(), a[k], rtol=1e-5, atol=1e-5), k


def test_train_unet_history_is_the_same_from_either_loader(tmp_path):
    from unet_amd import data as D
    from unet_amd.train import train_unet
    path = _npz(tmp_path, N=24)
    hist = {}
    for resident in (False, True):
        norms = dict(Te=D.MaskedLogStandardizer(eps=1.0), S=D.MaskedSymLogStandardizer(scale=5.0))
        tl, vl, norm, P, hw, scaler, pt, pk = D.make_loaders(path, batch_size=8, y_keys=["Te", "S"], norms_by_name=norms,
                                                             device="cuda", resident=resident)
        torch.manual_seed(5)
        _, h = train_unet(tl, vl, norm, 1, "cuda", out_ch=2, epochs=2, base=8, P=P, amp=False, multiscale=0,
                          save_path=str(tmp_path / f"m{int(resident)}.pt"), lam_grad_warmup_start=0, lam_grad_warmup_end=1)
        hist[resident] = h
    for k in ("tr_mse", "va_mse", "va_mae_eV"):
        assert np.allclose(hist[False][k], hist[True][k], rtol=2e-4, atol=1e-6), (k, hist[False][k], hist[True][k])
