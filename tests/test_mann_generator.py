"""Row f2: the Mann box generator of libwindgym_hip.so (wg_generate_mann_box: HIP spectral-tensor kernel + hipFFT) pinned
cell by cell against the numpy restatement fed with the IDENTICAL complex white noise, and the host-side eddy-lifetime
table against scipy's 2F1.  (hipersim itself — MannTurbulenceField.generate, Wind_Farm_Env.py:624-637 — is not
installable here: both sides restate the published algorithm, Mann 1998.)"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from windgym_amd import mann  # noqa: E402


def test_beta_table_matches_scipy_hyp2f1():
    """wg_mann_beta_table (Euler integral of 2F1(1/3, 17/6; 4/3; -x) by Gauss-Legendre panels, host code of the library)
    against scipy.special.hyp2f1 over the whole table range."""
    for gamma in (3.9, 2.0):
        tab = mann.mann_beta_table(gamma)
        kl = np.logspace(mann.BETA_TABLE["log10_lo"], mann.BETA_TABLE["log10_hi"], mann.BETA_TABLE["n"])
        ref = mann._eddy_lifetime_beta(kl, gamma)
        np.testing.assert_allclose(tab, ref, rtol=1e-12)
    assert np.all(mann.mann_beta_table(0.0) == 0.0)
    # asymptotes (Mann 1998): beta -> Gamma (kL)^(-2/3) for kL >> 1, ~ 1 / kL for kL << 1
    tab = mann.mann_beta_table(3.9)
    assert abs(tab[-1] / (3.9 * 1e6 ** (-2.0 / 3.0)) - 1.0) < 1e-6
    assert abs(tab[0] * 1e-6 / (tab[1] * 10 ** (-6 + 12 / 4095)) - 1.0) < 1e-6


def test_table_interpolation_is_a_faithful_beta():
    """the numpy restatement with the kernel's table interpolation == with 2F1 evaluated per cell (5e-4 of the field's
    standard deviation: the interpolation error of a 4096-point log table)"""
    rng = np.random.default_rng(0)
    shape = (3, 64, 32, 16)
    n = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) / np.sqrt(2.0)
    a = mann.mann_field_from_noise(n, (3.0, 3.0, 3.0))
    b = mann.mann_field_from_noise(n, (3.0, 3.0, 3.0), beta_table=mann.mann_beta_table(3.9))
    assert np.abs(a - b).max() < 5e-4
    assert abs(a[0].std() - 1.0) < 1e-12


def test_philox_noise_stream_of_the_generator():
    """wgo_mann_noise (the oracle-side restatement of the kernel's Philox stream): unit complex variance, independent
    parts, reproducible, different per seed"""
    from oracle import oracle as om
    n = om.mann_noise(7, (32, 16, 8))
    assert n.shape == (3, 32, 16, 8)
    assert abs(np.mean(np.abs(n) ** 2) - 1.0) < 0.03
    assert abs(n.real.var() - 0.5) < 0.02 and abs(n.imag.var() - 0.5) < 0.02
    assert abs(np.mean(n.real * n.imag)) < 0.02
    assert np.array_equal(n, om.mann_noise(7, (32, 16, 8)))
    assert not np.array_equal(n, om.mann_noise(8, (32, 16, 8)))


@pytest.mark.gpu
@pytest.mark.parametrize("dims,spacing,gamma,L", [
    ((256, 64, 32), (3.0, 3.0, 3.0), 3.9, 33.6),      # the sheared tensor of the ambient field
    ((256, 64, 32), (3.0, 3.0, 3.0), 0.0, 5.0),       # isotropic: the wake-added turbulence box (ADDED_BOX_SPEC)
    ((128, 48, 20), (4.0, 5.0, 6.0), 3.9, 33.6),      # non-cubic spacing, non-power-of-two dims
])
def test_hip_generator_equals_numpy_on_identical_noise(dims, spacing, gamma, L):
    """Same complex white noise into wg_generate_mann_box and into the float64 numpy restatement (with the kernel's beta
    table): every cell of the unit-variance box agrees to fp32 rounding."""
    rng = np.random.default_rng(11)
    shape = (3,) + dims
    n = ((rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) / np.sqrt(2.0)).astype(np.complex64)
    ref = mann.mann_field_from_noise(n.astype(np.complex128), spacing, 0.1, L, gamma, beta_table=mann.mann_beta_table(gamma))
    got = mann.generate_mann_box_hip(dims, spacing, 0.1, L, gamma, seed=0, noise=n).cpu().numpy()
    assert got.dtype == np.float32 and got.shape == shape
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4)
    assert abs(float(got[0].std()) - 1.0) < 1e-5


@pytest.mark.gpu
def test_hip_generator_builtin_noise_is_the_pinned_philox_stream():
    """Without a noise argument the kernel draws from Philox keyed by the seed: the numpy restatement fed with the
    oracle-side restatement of that stream reproduces the box (the float Box-Muller of the two sides may differ in the
    last bits of a few samples: 1e-3 of the field's standard deviation)."""
    from oracle import oracle as om
    dims, spacing = (128, 32, 16), (3.0, 3.0, 3.0)
    got = mann.generate_mann_box_hip(dims, spacing, seed=1234).cpu().numpy()
    ref = mann.mann_field_from_noise(om.mann_noise(1234, dims).astype(np.complex128), spacing,
                                     beta_table=mann.mann_beta_table(3.9))
    np.testing.assert_allclose(got, ref, rtol=1e-3, atol=1e-3)
    other = mann.generate_mann_box_hip(dims, spacing, seed=1235).cpu().numpy()
    assert np.abs(other - got).max() > 0.5            # another seed, another realisation


@pytest.mark.gpu
def test_hip_generator_reference_box_size_and_statistics():
    """the reference's MannFixed box (2048 x 512 x 64 @ 3 m, Wind_Farm_Env.py:649-658) generated on the device: unit
    variance of u, anisotropy of the sheared tensor (sigma_u > sigma_v > sigma_w), zero mean, a decaying correlation"""
    import torch
    spec = mann.reference_box_spec("MannFixed", 80.0)
    box = mann.generate_mann_box_hip(**spec)
    assert tuple(box.shape) == (3, 2048, 512, 64)
    sd = box.reshape(3, -1).std(dim=1).cpu().numpy()
    assert abs(sd[0] - 1.0) < 1e-4 and sd[0] > sd[1] > sd[2] > 0.4
    assert float(box.reshape(3, -1).mean(dim=1).abs().max()) < 1e-3
    u = box[0]
    c1 = float((u[:-8] * u[8:]).mean())        # 24 m along x
    c2 = float((u[:-64] * u[64:]).mean())      # 192 m
    assert 0.4 < c1 < 1.0 and c2 < c1
    assert torch.isfinite(box).all()
