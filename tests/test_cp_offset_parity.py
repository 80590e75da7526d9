"""GPU: the control-point offset stage (get_offset_image, MIMC_module.c:33-492) through the C ABI against the oracle
with the same shuffle seed.  Bar: same status, same offset, same control-point flags, same candidate / segment / CP
counts and BIT-identical vote sums (f32 sums of cluster means, i.e. every one of the 16 matches per candidate and the
clustering agree)."""
import numpy as np
import pytest

from mimc3_amd import synth

pytestmark = pytest.mark.gpu

K = [np.array([[-1, 0, 1]], np.float32), np.array([[-1], [0], [1]], np.float32),
     np.array([[-1 / 8] * 3, [-1 / 8, 1, -1 / 8], [-1 / 8] * 3], np.float32)]


@pytest.fixture(scope="module")
def api():
    from mimc3_amd import api as a
    return a


def cp_case(seed, shift, dimx=24, dimy=20, h=620, w=700, null_frac=0.0, slow=0.7, noise=0, tear=False):
    i0, i1 = synth.make_pair(h, w, shift, seed=seed, null_frac=null_frac, noise_dn=noise)
    xy = synth.make_grid(dimx, dimy, 60, 60, (w - 120) // dimx, (h - 120) // dimy, 1806.0, angle_deg=30.0)
    rng = np.random.default_rng(seed)
    s = rng.random(dimx * dimy) < slow
    xy[s, 4] = rng.uniform(-5, 5, s.sum()); xy[s, 5] = rng.uniform(-5, 5, s.sum())
    if tear:            # a region that moved differently: clusters disagree, more segments are needed
        i1 = i1.copy(); i1[:300, :350] = np.roll(i1[:300, :350], 3, axis=1)
    return i0, i1, xy


CASES = {
    "clean": dict(seed=1, shift=(2, -1)),
    "nulls_noise": dict(seed=2, shift=(-3, 4), null_frac=0.05, noise=2),
    "zero_shift_few_slow": dict(seed=3, shift=(0, 0), slow=0.2),
    "torn_heavy_nulls": dict(seed=6, shift=(-1, 1), null_frac=0.5, noise=40, slow=0.9, tear=True),
    "torn_noise": dict(seed=7, shift=(2, 2), noise=60, slow=0.5, tear=True),
}


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("seed", [3, 99])
def test_vs_oracle(api, oracle, name, seed):
    i0, i1, xy = cp_case(**CASES[name])
    rc, off, flag, info, sduv = oracle.get_offset_image(i0, i1, xy, K, seed, num_cp_min=20)
    with api.Context(0) as ctx:
        ctx.set_images(i0, i1)
        st, o2, f2, info2, sduv2 = ctx.get_offset_image(xy, K, seed=seed, num_cp_min=20)
    assert st == rc
    assert np.array_equal(info2, info), (info2, info)
    assert np.array_equal(f2, flag)
    assert np.array_equal(sduv2.view(np.uint32), sduv.view(np.uint32)), (sduv2, sduv)
    assert np.array_equal(o2, off)
    if name == "clean":
        assert st == 1 and tuple(off) == CASES[name]["shift"]


@pytest.mark.parametrize("name,peers", [("clean", 1), ("torn_heavy_nulls", 2), ("torn_noise", 3), ("nulls_noise", 7)])
def test_slices_over_several_contexts(api, oracle, name, peers):
    """mimc3_get_offset_image_multi: every segment's candidates cut into slices matched by 1 + peers contexts on host threads
    of their own (here all on the one device): status, offset, flags, counts and the f32 vote sums of the oracle, as with one."""
    i0, i1, xy = cp_case(**CASES[name])
    rc, off, flag, info, sduv = oracle.get_offset_image(i0, i1, xy, K, 5, num_cp_min=20)
    ctxs = [api.Context(0) for _ in range(1 + peers)]
    try:
        for c in ctxs:
            c.set_images(i0, i1)
        st, o2, f2, info2, sduv2 = ctxs[0].get_offset_image(xy, K, seed=5, num_cp_min=20, peers=ctxs[1:])
        # a context listed twice is refused (two host threads would slice on the same scratch and streams)
        with pytest.raises(api.Mimc3Error):
            ctxs[0].get_offset_image(xy, K, seed=5, num_cp_min=20, peers=[ctxs[1], ctxs[1]] if peers > 1 else [ctxs[0]])
        # a peer that holds another pair is refused, not used
        ctxs[-1].set_images(i0[:-8], i1[:-8])
        with pytest.raises(api.Mimc3Error):
            ctxs[0].get_offset_image(xy, K, seed=5, num_cp_min=20, peers=ctxs[1:])
    finally:
        for c in ctxs:
            c.close()
    assert st == rc
    assert np.array_equal(info2, info), (info2, info)
    assert np.array_equal(f2, flag)
    assert np.array_equal(sduv2.view(np.uint32), sduv.view(np.uint32)), (sduv2, sduv)
    assert np.array_equal(o2, off)


@pytest.mark.parametrize("seed", [3, 99])
def test_16bit_pair_vs_oracle(api, oracle, seed):
    """16-bit DN: the atlases of all four image variants take the register-tiled f32 kernel (16 concurrent matches on four
    streams sharing each variant's f32 planes).  Integer DN keeps every f64 sum exact in any order, so votes are bit-identical."""
    i0, i1 = synth.make_pair(620, 700, (2, -3), seed=31, null_frac=0.03, noise_dn=300, bits=16)
    xy = synth.make_grid(24, 20, 60, 60, 24, 25, 1806.0, angle_deg=30.0)
    rng = np.random.default_rng(5)
    s = rng.random(xy.shape[0]) < 0.7
    xy[s, 4] = rng.uniform(-5, 5, s.sum()); xy[s, 5] = rng.uniform(-5, 5, s.sum())
    rc, off, flag, info, sduv = oracle.get_offset_image(i0, i1, xy, K, seed, num_cp_min=20)
    with api.Context(0) as ctx:
        ctx.set_images(i0, i1)
        for _ in range(2):
            st, o2, f2, info2, sduv2 = ctx.get_offset_image(xy, K, seed=seed, num_cp_min=20)
            assert st == rc == 1
            assert np.array_equal(info2, info), (info2, info)
            assert np.array_equal(f2, flag)
            assert np.array_equal(sduv2.view(np.uint32), sduv.view(np.uint32)), (sduv2, sduv)
            assert np.array_equal(o2, off)


def test_not_enough_candidates(api, oracle):
    i0, i1, xy = cp_case(seed=4, shift=(5, 5), slow=0.05)
    rc, off, flag, info, _ = oracle.get_offset_image(i0, i1, xy, K, 5, num_cp_min=20)
    with api.Context(0) as ctx:
        ctx.set_images(i0, i1)
        st, o2, f2, info2, _ = ctx.get_offset_image(xy, K, seed=5, num_cp_min=20)
    assert rc == -1 and st == -1 and not f2.any() and np.array_equal(info2, info)


def test_filter_state_is_ignored_and_bounds(api, oracle):
    i0, i1, xy = cp_case(seed=1, shift=(2, -1))
    with api.Context(0) as ctx:
        ctx.set_images(i0, i1)
        a = ctx.get_offset_image(xy, K, seed=8, num_cp_min=20)
        ctx.filter_images(K[2])
        b = ctx.get_offset_image(xy, K, seed=8, num_cp_min=20)       # still works on the pair as handed over
        assert a[0] == b[0] == 1 and np.array_equal(a[1], b[1]) and np.array_equal(a[4].view(np.uint32), b[4].view(np.uint32))
        xy2 = xy.copy(); xy2[0, 2] = 20.0; xy2[0, 4] = xy2[0, 5] = 0.0   # a slow point 20 px from the border
        with pytest.raises(api.Mimc3Error):
            ctx.get_offset_image(xy2, K, seed=8, num_cp_min=20)


def test_golden(api):
    """reference-generated fixture (tests/golden/cp_small.npz): status, offset, control-point flags"""
    from test_oracle_golden import _cp_golden
    z, par, ks = _cp_golden()
    with api.Context(0) as ctx:
        ctx.set_images(z["i0"].astype(np.float32), z["i1"].astype(np.float32))
        for sd in z["seeds"]:
            st, off, flag, info, _ = ctx.get_offset_image(z["xyuvav"], ks, seed=int(sd), **par)
            assert st == int(z[f"rc_{sd}"]) and np.array_equal(off, z[f"offset_{sd}"]) and np.array_equal(flag, z[f"flag_{sd}"])
