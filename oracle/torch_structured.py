"""
TEST / BASELINE INFRASTRUCTURE ONLY -- never imported by the product (adorym_amd/).

A PyTorch-CPU *autograd* restatement of the reference's hot path with the reference's OP STRUCTURE
(CPU baseline flavour ii): real and imaginary parts carried as separate real tensors,
one strided [..., i_slice, c] select per slice, cos/sin/exp modulator, re-packing to complex around every
FFT, torch.autograd.grad for the whole backward.  It exists so that bench.py can time "what the reference
does on a CPU" on the GPU box, where the reference itself cannot travel.  Written from scratch from the
behaviour of
    adorym/forward_model.py:264-353   (pad, tile gather, stack)
    adorym/propagate.py:195-270       (modulate / convolve loop, far field)
    adorym/wrappers.py:600-608,699-739,774-779,796-813,1062-1067
    adorym/forward_model.py:88-103    (LSQ loss)
and checked against the NumPy oracle (tests/test_oracle_vs_golden.py::test_torch_structured_matches_oracle)
and, in the build container, timed against the real reference (oracle/time_vs_reference.py).
"""
import numpy as np
import torch


def _cmul(ar, ai, br, bi):
    return ar * br - ai * bi, ar * bi + ai * br


def _fft2(re, im, inverse=False):
    z = torch.complex(re, im)
    z = torch.fft.ifft2(z, dim=(1, 2)) if inverse else torch.fft.fft2(z, dim=(1, 2))
    return z.real, z.imag


def propagate_batch(tiles, probe_re, probe_im, h_re, h_im, k1, far_field=True, sign=1):
    """tiles [B,P,P,S,2] (delta, beta) -> exit wave (re, im) [B,P,P] at the detector."""
    S = tiles.shape[3]
    wr, wi = probe_re, probe_im
    for s in range(S):
        delta = tiles[:, :, :, s, 0]
        beta = tiles[:, :, :, s, 1]
        mag = torch.exp(-k1 * beta)
        ph = -sign * k1 * delta
        wr, wi = _cmul(wr, wi, mag * torch.cos(ph), mag * torch.sin(ph))
        if s < S - 1:
            fr, fi = _fft2(wr, wi)
            fr, fi = _cmul(fr, fi, h_re, h_im)
            wr, wi = _fft2(fr, fi, inverse=True)
    if far_field:
        wr, wi = _fft2(wr, wi, inverse=(sign != 1))
        wr = torch.fft.fftshift(wr, dim=(1, 2))
        wi = torch.fft.fftshift(wi, dim=(1, 2))
    return wr, wi


def loss_and_grad(obj_rot, pos, probe, h, k1, meas, far_field=True, sign=1, dtype=torch.float32):
    """obj_rot [Y,X,Z,2] numpy; pos [B,2] int; probe [P,P] complex; h [P,P] complex; meas [B,P,P] magnitudes.
    Returns (loss, d loss / d obj_rot) with the loss = mean((|psi| - meas)^2)."""
    P = probe.shape[0]
    obj = torch.tensor(obj_rot, dtype=dtype, requires_grad=True)
    Y, X = obj.shape[:2]
    pos = np.asarray(pos, dtype=int)
    py0, px0 = max(0, -pos[:, 0].min()), max(0, -pos[:, 1].min())
    py1, px1 = max(0, pos[:, 0].max() + P - Y), max(0, pos[:, 1].max() + P - X)
    padded = torch.nn.functional.pad(obj, (0, 0, 0, 0, px0, px1, py0, py1))
    tiles = torch.stack([padded[y + py0:y + py0 + P, x + px0:x + px0 + P] for y, x in pos])
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=dtype)
    er, ei = propagate_batch(tiles, t(probe.real), t(probe.imag), t(h.real), t(h.imag), k1, far_field, sign)
    pred = torch.norm(torch.stack([er, ei], 0), dim=0)
    loss = torch.mean((pred - t(meas)) ** 2)
    g, = torch.autograd.grad(loss, [obj])
    return float(loss.detach()), g.numpy()


# run as a script: python oracle/torch_structured.py in.npz out.npz  (oracle/torch_child.py drives it from a torch-free process)
if __name__ == '__main__':
    import sys
    f_ = np.load(sys.argv[1])
    if int(f_['threads']) > 0:
        torch.set_num_threads(int(f_['threads']))
    loss_, grad_ = loss_and_grad(f_['obj_rot'], f_['pos'], f_['probe'], f_['h'], float(f_['k1']), f_['meas'])
    np.savez(sys.argv[2], loss=np.float64(loss_), grad=grad_)
