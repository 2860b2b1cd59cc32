"""
Deterministic input builders shared by the golden generator (gen_goldens.py, runs only in the
development container where /root/reference exists) and by the parity tests (run anywhere).
Fixtures (*.npz) therefore hold reference OUTPUTS plus the few inputs that are themselves
produced by the reference-side setup; everything else is rebuilt from these seeds.
"""
import numpy as np

ENERGY_EV = 5000.
PSIZE_CM = 1e-7


def rng(seed):
    return np.random.default_rng(seed)


def smooth_field(shape, seed, cutoff=0.25):
    """Band-limited random field in [0, 1] (structured object / initial guess)."""
    r = rng(seed)
    a = r.standard_normal(shape)
    f = np.fft.fftn(a)
    grids = np.meshgrid(*[np.fft.fftfreq(n) for n in shape], indexing='ij')
    rad = np.sqrt(sum(g ** 2 for g in grids))
    f = f * (rad <= cutoff)
    a = np.real(np.fft.ifftn(f))
    a = a - a.min()
    return a / a.max()


# ---------------------------------------------------------------- F2 / F3 tile-level cases
# name: (P, S, free_prop_cm, sigma, binning, n_modes, probe kind, normalize_fft, fresnel_approx, dscale)
TILE_CASES = {
    'p12_s9_far_pos':      (12, 9, 'inf', 1, 1, 1, 'random', False, True, 2e-3),
    'p12_s9_far_neg':      (12, 9, 'inf', -1, 1, 1, 'random', False, True, 2e-3),
    'p12_s9_near':         (12, 9, 0, 1, 1, 1, 'random', False, True, 2e-3),
    'p12_s9_fresnel':      (12, 9, 1e-4, 1, 1, 1, 'random', False, True, 2e-3),
    'p16_s32_far_bin4':    (16, 32, 'inf', 1, 4, 1, 'random', False, True, 2e-3),
    'p16_s10_far_bin4':    (16, 10, 'inf', 1, 4, 1, 'random', False, True, 2e-3),
    'p12_s9_far_modes3':   (12, 9, 'inf', 1, 1, 3, 'random', False, True, 2e-3),
    'p12_s1_far':          (12, 1, 'inf', 1, 1, 1, 'random', False, True, 2e-3),
    'p12_s9_far_ortho':    (12, 9, 'inf', 1, 1, 1, 'random', True, True, 2e-3),
    'p12_s9_far_nofresnel': (12, 9, 'inf', 1, 1, 1, 'random', False, False, 2e-3),
    'p64_s8_near_plane':   (64, 8, 0, 1, 1, 1, 'plane', False, True, 2e-3),
    'p72_s9_far_gauss':    (72, 9, 'inf', 1, 1, 1, 'gaussian', False, True, 2e-3),
    'p72_s9_far_random':   (72, 9, 'inf', 1, 1, 1, 'random', False, True, 2e-3),
}
TILE_B = 3


def tile_case_inputs(name):
    P, S, fp, sigma, binning, n_modes, pkind, norm, fa, dscale = TILE_CASES[name]
    seed = abs(hash_name(name)) % (2 ** 31)
    r = rng(seed)
    shape = (TILE_B, P, P, S)
    # guess != truth; strong object (delta ~ 2e-3, beta ~ 2e-4) => total phase O(1 rad), well conditioned
    guess = np.stack([r.uniform(0, dscale, shape), r.uniform(0, dscale * 0.1, shape)], -1)
    truth = np.stack([r.uniform(0, dscale, shape), r.uniform(0, dscale * 0.1, shape)], -1)
    probes = []
    for m in range(n_modes):
        if pkind == 'random':
            mag = 0.5 + r.uniform(0, 1, (P, P))
            ph = r.uniform(-np.pi, np.pi, (P, P))
            probes.append(mag * np.exp(1j * ph))
        elif pkind == 'plane':
            probes.append(np.ones((P, P), dtype=complex))
        else:
            py = np.arange(P) - (P - 1.) / 2
            xx, yy = np.meshgrid(py, py)
            mag = np.exp(-(xx ** 2 + yy ** 2) / (2 * 6. ** 2))
            ph = 0.5 * np.exp(-(xx ** 2 + yy ** 2) / (2 * 6. ** 2))
            probes.append(mag * np.exp(1j * ph))
    probes = np.stack(probes)
    return dict(P=P, S=S, free_prop_cm=fp, sigma=sigma, binning=binning, n_modes=n_modes,
                normalize_fft=norm, fresnel_approx=fa, guess=guess, truth=truth, probes=probes)


def hash_name(name):
    h = 0
    for ch in name:
        h = (h * 131 + ord(ch)) % 1000003
    return h


# ---------------------------------------------------------------- F4 rotation cases
ROT_CASES = {
    'n12_t0.3':     ((5, 12, 12), 0.3),
    'n12_halfpi':   ((5, 12, 12), np.pi / 2),
    'n12_t3.44159': ((5, 12, 12), 3.44159),
    'n16_twopi':    ((4, 16, 16), 2 * np.pi),
    'n16_t1.0':     ((4, 16, 16), 1.0),
}


def rot_case_inputs(name):
    size, theta = ROT_CASES[name]
    r = rng(hash_name(name))
    obj = 0.5 + r.standard_normal(size + (2,))
    cot = r.standard_normal(size + (2,))
    return size, np.float32(theta), obj, cot


# ---------------------------------------------------------------- F6 end-to-end driver case
E2E = dict(N=32, P=16, n_theta=4, grid=3, step=8, origin=-4, energy_ev=ENERGY_EV, psize_cm=PSIZE_CM,
           minibatch_size=3)


def e2e_inputs():
    N = E2E['N']
    truth_d = 1.0e-3 * smooth_field((N, N, N), 11)
    truth_b = 1.0e-4 * smooth_field((N, N, N), 12)
    # structured initial guess: blurred truth * 0.5 + small noise
    guess_d = 0.5e-3 * smooth_field((N, N, N), 11, cutoff=0.1) + 1e-6 * rng(13).standard_normal((N, N, N))
    guess_b = 0.5e-4 * smooth_field((N, N, N), 12, cutoff=0.1) + 1e-7 * rng(14).standard_normal((N, N, N))
    g = E2E['grid']
    ys = np.arange(g) * E2E['step'] + E2E['origin']
    probe_pos = np.array([(y, x) for y in ys for x in ys], dtype=float)
    theta_ls = np.linspace(0, 2 * np.pi, E2E['n_theta'], dtype='float32')
    P = E2E['P']
    py = np.arange(P) - (P - 1.) / 2
    xx, yy = np.meshgrid(py, py)
    r = rng(15)
    mag = np.exp(-(xx ** 2 + yy ** 2) / (2 * 5. ** 2))
    ph = r.uniform(-np.pi, np.pi, (P, P))          # random-phase probe => well-conditioned far field
    return dict(truth=(truth_d, truth_b), guess=(guess_d, guess_b), probe_pos=probe_pos, theta_ls=theta_ls,
                probe_mag=mag, probe_phase=ph)


# ---------------------------------------------------------------- F14: the reference driver at world size 2 (config 4's exchange)
# name: (scan positions used, keyword arguments).  9 positions, minibatch 3, 2 ranks: a global batch of 6 can straddle two
# angles and the reference's per-rank optimiser counters drift apart ('immediate' records that); with 6 positions they cannot.
W2_RUNS = {
    'immediate':      (9, dict(n_epochs=1, optimizer='adam', learning_rate=1e-6)),
    'immediate6':     (6, dict(n_epochs=2, optimizer='adam', learning_rate=1e-6)),
    'immediate6_reg': (6, dict(n_epochs=1, optimizer='adam', learning_rate=1e-6, gamma=1e-6, alpha_d=1e-4, alpha_b=1e-5)),
    'perangle':       (9, dict(n_epochs=1, optimizer='adam', learning_rate=1e-6, update_scheme='per angle')),
    'probe6':         (6, dict(n_epochs=2, optimizer='adam', learning_rate=1e-6, optimize_probe=True, probe_learning_rate=1e-3)),
}


# ---------------------------------------------------------------- F17: config-3 depth (P = 72, 256 slices) on a small lateral object
def depth256_inputs():
    r = rng(41)
    P, S, Y, X = 72, 256, 84, 84
    obj = np.stack([3e-4 * smooth_field((Y, X, S), 7, cutoff=0.15), 1.5e-5 * smooth_field((Y, X, S), 8, cutoff=0.15)], -1)
    pos = np.array([(0, 0), (12, 12), (-6, 5)])
    probe = (0.5 + r.uniform(0, 1, (P, P))) * np.exp(1j * r.uniform(-np.pi, np.pi, (P, P)))
    truth = np.stack([3e-4 * smooth_field((Y, X, S), 17, cutoff=0.15), 1.5e-5 * smooth_field((Y, X, S), 18, cutoff=0.15)], -1)
    return dict(P=P, S=S, obj=obj, pos=pos, probe=probe, truth=truth)


# ---------------------------------------------------------------- F15: rotate_out_of_loop through the driver (E2E inputs)
ROOL_RUNS = {
    'immediate': dict(n_epochs=2, optimizer='adam', learning_rate=1e-6),
    'immediate_reg': dict(n_epochs=1, optimizer='adam', learning_rate=1e-6, gamma=1e-6, alpha_d=1e-4, alpha_b=1e-5),
    'perangle': dict(n_epochs=1, optimizer='adam', learning_rate=1e-6, update_scheme='per angle'),
}


# ---------------------------------------------------------------- F11: config-1-shaped 2-D ptychography (f2 row)
C1MINI = dict(Y=40, X=44, P=16, M=2, energy_ev=8801.121930115722, psize_cm=1.32789376566526e-06, minibatch_size=5, n_dp_batch=2)


def c1mini_inputs():
    """2-D complex-transmission object, two incoherent probe modes, raster scan with sub-pixel positions."""
    c = C1MINI
    Y, X, P, M = c['Y'], c['X'], c['P'], c['M']
    mag_t = 1 - 0.35 * smooth_field((Y, X, 1), 111)
    ph_t = 0.9 * smooth_field((Y, X, 1), 112) - 0.4
    guess_mag = np.full((Y, X, 1), 0.85) + 0.02 * smooth_field((Y, X, 1), 113)
    guess_ph = 0.05 * smooth_field((Y, X, 1), 114)
    r = rng(115)
    grid = np.array([(y, x) for y in range(-5, 32, 7) for x in range(-6, 36, 8)], dtype=float)
    pos_true = grid + r.uniform(-0.45, 0.45, grid.shape)            # where the data were taken
    pos_nominal = grid + r.uniform(-0.3, 0.3, grid.shape)           # what the reconstruction is told (non-integer)
    py = np.arange(P) - (P - 1.) / 2
    xx, yy = np.meshgrid(py, py)
    env = np.exp(-(xx ** 2 + yy ** 2) / (2 * 4. ** 2))
    pm = np.stack([env * (0.6 + 0.4 * r.uniform(size=(P, P))) * (1.0 if m == 0 else 0.35) for m in range(M)])
    pp = np.stack([r.uniform(-np.pi, np.pi, (P, P)) for m in range(M)])
    guess_pm = pm * (1 + 0.1 * r.uniform(-1, 1, pm.shape))
    guess_pp = pp + 0.1 * r.uniform(-1, 1, pp.shape)
    return dict(truth=(mag_t, ph_t), guess=(guess_mag, guess_ph), pos_true=pos_true, pos_nominal=pos_nominal,
                probe_true=(pm, pp), probe_guess=(guess_pm, guess_pp))


# ---------------------------------------------------------------- F12: multi-distance holography (f1 row, config-5 shape)
C5MINI = dict(N=32, energy_ev=17050., psize_cm=1e-4, dists_cm=(40., 60., 90.), )


def c5mini_inputs():
    """2-D complex-transmission object seen by a plane wave at three propagation distances; the holograms of
    distances 1 and 2 are recorded with a small affine misregistration."""
    c = C5MINI
    N = c['N']
    mag_t = 1 - 0.25 * smooth_field((N, N, 1), 121)
    ph_t = 0.6 * smooth_field((N, N, 1), 122) - 0.3
    guess_mag = np.full((N, N, 1), 0.9) + 0.02 * smooth_field((N, N, 1), 123)
    guess_ph = 0.05 * smooth_field((N, N, 1), 124)
    affine_true = np.array([[[1., 0, 0], [0, 1., 0]], [[1.02, 0.01, 0.03], [-0.01, 0.99, -0.02]], [[0.98, 0.0, -0.04], [0.015, 1.01, 0.02]]])
    dists_guess = np.array(c['dists_cm']) * np.array([1.0, 1.03, 0.97])
    return dict(truth=(mag_t, ph_t), guess=(guess_mag, guess_ph), affine_true=affine_true, dists_guess=dists_guess)


# ---------------------------------------------------------------- f1 divided into sub-tiles with a safe zone (golden F18)
C5TILES = dict(N=48, SUB=16, energy_ev=17050., psize_cm=1e-4, dists_cm=(40., 60., 90.), minibatch_size=4, n_epochs=2, learning_rate=1e-2,
               runs={'ri_szw4': dict(unknown_type='real_imag', szw=4, probe='plane'),
                     'db_szw4_probe': dict(unknown_type='delta_beta', szw=4, probe='supplied'),
                     'ri_szw0': dict(unknown_type='real_imag', szw=0, probe='plane')})


def c5tiles_inputs(run):
    """A 48 x 48 one-slice object recorded at three distances as 3 x 3 holograms of 16 x 16 pixels each (n_blocks = 9); the
    reconstruction propagates every tile with a safe zone of ``szw`` pixels around it.  'supplied': a probe that varies over the
    field of view (real and imaginary part), so that every tile sees its own window of it."""
    c = C5TILES
    N, SUB = c['N'], c['SUB']
    r = c['runs'][run]
    pos = np.array([[y, x] for y in range(0, N, SUB) for x in range(0, N, SUB)], dtype=float)
    if r['unknown_type'] == 'real_imag':
        mag_t = 1 - 0.25 * smooth_field((N, N, 1), 321)
        ph_t = 0.6 * smooth_field((N, N, 1), 322) - 0.3
        truth = np.stack([mag_t * np.cos(ph_t), mag_t * np.sin(ph_t)], -1)
        guess = (np.full((N, N, 1), 0.9) + 0.02 * smooth_field((N, N, 1), 323), 0.05 * smooth_field((N, N, 1), 324))   # (mag, phase)
    else:
        truth = np.stack([2e-6 * smooth_field((N, N, 1), 331), 2e-7 * smooth_field((N, N, 1), 332)], -1)
        guess = (5e-7 * (1 + smooth_field((N, N, 1), 333)), 5e-8 * (1 + smooth_field((N, N, 1), 334)))                # (delta, beta)
    if r['probe'] == 'supplied':
        pm = 1 + 0.2 * smooth_field((N, N, 1), 341)[..., 0]
        pp = 0.5 * smooth_field((N, N, 1), 342)[..., 0] - 0.25
    else:
        pm, pp = np.ones((N, N)), np.zeros((N, N))
    return dict(pos=pos, truth=truth, guess=guess, probe_mag=pm, probe_phase=pp, szw=r['szw'], unknown_type=r['unknown_type'],
                probe_type=r['probe'])


# ---------------------------------------------------------------- full-size config 3 through the driver (tests/test_gpu_fullsize.py)
FULLSIZE = dict(N=256, P=72, rows=(10, 13), theta=0.4, margin=4)


def smooth_field_fast(shape, seed, cutoff=0.12):
    """Band-limited random field in [0, 1] like smooth_field(), synthesised directly in the (half) spectrum: one inverse real
    transform instead of a forward and an inverse complex one -- 256^3 in about a second."""
    import scipy.fft as sfft
    r = rng(seed)
    f = np.zeros(shape[:-1] + (shape[-1] // 2 + 1,), dtype=np.complex64)
    k = [max(1, int(np.ceil(cutoff * n))) for n in shape]
    grids = np.meshgrid(np.fft.fftfreq(shape[0])[np.r_[0:k[0] + 1, -k[0]:0]], np.fft.fftfreq(shape[1])[np.r_[0:k[1] + 1, -k[1]:0]],
                        np.fft.rfftfreq(shape[2])[:k[2] + 1], indexing='ij')
    keep = np.sqrt(sum(g ** 2 for g in grids)) <= cutoff
    blk = (r.standard_normal(keep.shape) + 1j * r.standard_normal(keep.shape)) * keep
    f[np.ix_(np.r_[0:k[0] + 1, -k[0]:0], np.r_[0:k[1] + 1, -k[1]:0], np.arange(k[2] + 1))] = blk.astype(np.complex64)
    a = sfft.irfftn(f, s=shape, workers=-1).astype(np.float64)
    a -= a.min()
    return a / a.max()


# TWO angles (0.4 rad and a 45-degree-class one), 64 positions = two full minibatches per angle with DIFFERENT y-footprints (rows
# 10-11, then the rest of row 11 with 18 positions of row 13): four 'immediate' minibatches, the transmission cache refilled and
# the adjoint CSR rebuilt for the new angle, the optimiser's step counter advancing at the angle boundary
# (adorym/ptychography.py:1266-1271); four updates -> the TV stencil reaches four planes into the margin.  (Six updates at
# this learning rate put even the oracle's own fp32 run beyond BASELINE's absolute RMSE bound of 1e-5: Adam's early steps are
# lr * sign(g), and the count of rounding-level sign flips grows with every update.)
FULLSIZE2 = dict(N=256, P=72, rows=(10, 12), extra=(13, 18), thetas=(0.4, 0.78), margin=5)


def fullsize_inputs(case=1):
    """BASELINE config 3 at its own size -- 256^3 object, 72 x 72 probe, the 23 x 23 scan with 12-pixel steps, minibatch 32,
    L1 + TV, Adam with the configuration's learning rate -- on three rows of the scan (69 positions = three minibatches of 32
    after the reference's random top-up) at one angle.  The object is smooth and non-zero EVERYWHERE (the deferred part of the
    split Adam pass works on all planes).  A rotation about axis 0 never mixes y planes, so the CPU checker works on the slab of
    planes the positions touch plus `margin` planes either side (the TV stencil reaches one plane further with every update);
    the measured data are the fp64 forward model of a second smooth object on that slab."""
    F = FULLSIZE if case == 1 else FULLSIZE2
    N, P = F['N'], F['P']
    ys = np.arange(23) * 12 - 36
    allpos = np.array([(y, x) for y in ys for x in ys], dtype=float)
    pos = allpos[F['rows'][0] * 23:F['rows'][1] * 23]
    if 'extra' in F:
        pos = np.concatenate([pos, allpos[F['extra'][0] * 23:F['extra'][0] * 23 + F['extra'][1]]])
    y_lo, y_hi = int(pos[:, 0].min()), int(pos[:, 0].max()) + P
    s0, s1 = y_lo - F['margin'], y_hi + F['margin']
    guess = np.stack([3e-4 * smooth_field_fast((N, N, N), 271), 1.5e-5 * smooth_field_fast((N, N, N), 272)], -1)
    shp = (s1 - s0, N, N)
    # (the truth is related to the guess, as in a reconstruction under way: the data term then pulls on every footprint voxel)
    truth_slab = np.stack([3e-4 * smooth_field_fast(shp, 275), 1.5e-5 * smooth_field_fast(shp, 276)], -1)
    truth_slab = 0.6 * guess[s0:s1] + 0.4 * truth_slab
    thetas = np.linspace(F['thetas'][0], F['thetas'][1], 2, dtype='float32') if case != 1 else np.array([F['theta']], dtype='float32')
    return dict(pos=pos, y_lo=y_lo, y_hi=y_hi, s0=s0, s1=s1, guess=guess, truth_slab=truth_slab, theta=thetas[0], thetas=thetas)
