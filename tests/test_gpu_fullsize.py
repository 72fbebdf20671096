"""The headline kernels at the benchmark's FULL batch size (1080p CTU grid x 128 frames: 4 147 200 8x8 pairs, 253 440 32x32
blocks -- bench.py's workload): EVERY output of the launch bench.py times is compared bit for bit with the oracle (its C
loops run over the host cores, a few seconds per launch), next to size-independent properties (symmetry / identity, sums
against an independent torch computation, the forward-inverse round trip).  The same for one 4K frame's worth of blocks
(BASELINE.json configs[4]: 3840x2160) of every SAD / SATD size and every transform, with ragged counts, so that every
grid-cap and tail branch of the launch code is taken at counts far beyond the small parity cases."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
import ref_lib as R

pytestmark = pytest.mark.gpu

FRAMES = 128
N8 = 32400 * FRAMES
N32 = 1980 * FRAMES


@pytest.fixture(scope="module")
def env():
    import torch
    from kvazaar_amd import _lib
    L = _lib.init(0)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(4242)
    return torch, _lib, L, dev, g


def test_sad_satd_8x8_full_batch(env):
    torch, _lib, L, dev, g = env
    cur = torch.randint(0, 256, (N8, 64), dtype=torch.uint8, device=dev, generator=g)
    noise = torch.randint(-12, 13, (N8, 64), dtype=torch.int16, device=dev, generator=g)
    ref = (cur.to(torch.int16) + noise).clamp_(0, 255).to(torch.uint8)
    del noise
    out = [torch.empty(N8, dtype=torch.int32, device=dev) for _ in range(4)]
    torch.cuda.synchronize()                                  # the library's stream does not wait for torch's: inputs must be complete
    _lib.check(L.kvz_hip_sad_nxn_batch(8, cur.data_ptr(), ref.data_ptr(), N8, out[0].data_ptr(), None), "sad")
    _lib.check(L.kvz_hip_sad_nxn_batch(8, ref.data_ptr(), cur.data_ptr(), N8, out[1].data_ptr(), None), "sad swapped")
    _lib.check(L.kvz_hip_satd_nxn_batch(8, cur.data_ptr(), ref.data_ptr(), N8, out[2].data_ptr(), None), "satd")
    _lib.check(L.kvz_hip_satd_nxn_batch(8, ref.data_ptr(), cur.data_ptr(), N8, out[3].data_ptr(), None), "satd swapped")
    _lib.check(L.kvz_hip_stream_sync(None), "sync")
    sad, sad_sw, satd, satd_sw = out
    # symmetry
    assert bool((sad == sad_sw).all()) and bool((satd == satd_sw).all())
    # every SAD against an independent computation (torch), chunked to bound memory
    for lo in range(0, N8, 1 << 20):
        hi = min(N8, lo + (1 << 20))
        want = (cur[lo:hi].to(torch.int16) - ref[lo:hi].to(torch.int16)).abs().sum(dim=1, dtype=torch.int32)
        assert bool((sad[lo:hi] == want).all()), "sad chunk at %d" % lo
    # the Hadamard transform preserves energy: 64 * sum(d^2) = sum(coef^2), so (sum |coef|)^2 >= 64 sum d^2 >= ... gives
    # sad / 8 <= (satd * 4 + 2) / 8 ... use the two sure bounds: satd == 0 iff the blocks are equal, and sum|coef| >= |DC|
    same = (cur == ref).all(dim=1)
    assert bool((satd[same] == 0).all()) and bool((satd[~same] > 0).all())
    dc = (cur.to(torch.int32).sum(dim=1) - ref.to(torch.int32).sum(dim=1)).abs()
    assert bool(((satd.to(torch.int64) * 4 + 2) >= dc.to(torch.int64)).all())
    # identity
    _lib.check(L.kvz_hip_satd_nxn_batch(8, cur.data_ptr(), cur.data_ptr(), N8, out[3].data_ptr(), None), "satd self")
    _lib.check(L.kvz_hip_stream_sync(None), "sync")
    assert int(out[3].abs().max()) == 0
    # exact against the oracle, every one of the 4 147 200 results of the launch the benchmark times
    a, b = cur.cpu().numpy(), ref.cpu().numpy()
    np.testing.assert_array_equal(sad.cpu().numpy().view(np.uint32), O.cost_nxn_many("sad", 8, a, b))
    np.testing.assert_array_equal(satd.cpu().numpy().view(np.uint32), O.cost_nxn_many("satd", 8, a, b))
    if R.available():                                           # and against the reference's own generic strategy
        np.testing.assert_array_equal(satd.cpu().numpy().view(np.uint32), R.cost_nxn_many("satd", 8, a, b))


def test_dct_32x32_full_batch(env):
    torch, _lib, L, dev, g = env
    res = torch.randint(-255, 256, (N32, 1024), dtype=torch.int16, device=dev, generator=g)
    res[::7] = 0                                              # zero blocks stay zero
    res[1::7] = res[1::7, :1].expand(-1, 1024)                # flat blocks: only the DC coefficient, 128 * value
    coef = torch.empty_like(res)
    back = torch.empty_like(res)
    torch.cuda.synchronize()                                  # the library's stream does not wait for torch's: inputs must be complete
    _lib.check(L.kvz_hip_transform_batch(0, 32, res.data_ptr(), coef.data_ptr(), N32, None), "dct")
    _lib.check(L.kvz_hip_transform_batch(1, 32, coef.data_ptr(), back.data_ptr(), N32, None), "idct")
    _lib.check(L.kvz_hip_stream_sync(None), "sync")
    assert int(coef[::7].abs().max()) == 0
    flat = coef[1::7]
    assert bool((flat[:, 0].to(torch.int32) == 128 * res[1::7, 0].to(torch.int32)).all()) and int(flat[:, 1:].abs().max()) == 0
    # forward + inverse returns the residual up to the rounding of HEVC's integer transform pair: the reference's own
    # generic pair has a mean round-trip error of 0.78 and a maximum of 5 over 4 M samples of this distribution
    # (6 over this batch's 260 M) -- a transposed, shifted or mis-scaled block would be off by hundreds
    err = (back.to(torch.int32) - res.to(torch.int32)).abs()
    assert int(err.max()) <= 8, int(err.max())
    assert float(err.to(torch.float32).mean()) < 0.85
    # exact against the oracle: all 253 440 blocks of both launches
    x, c = res.cpu().numpy(), coef.cpu().numpy()
    np.testing.assert_array_equal(c, O.transform_many("dct", 32, x))
    np.testing.assert_array_equal(back.cpu().numpy(), O.transform_many("idct", 32, c))
    if R.available():
        np.testing.assert_array_equal(c, R.transform_many("dct", 32, x))


def _frame_4k_count(n):
    return (3840 // n) * (2160 // n)


@pytest.mark.parametrize("n", [4, 8, 16, 32, 64])
def test_sad_satd_4k_frame_every_size(env, n):
    """one 3840x2160 frame's worth of n x n block pairs (+ a ragged tail of 13 blocks), plain and dual, every result exact"""
    torch, _lib, L, dev, g = env
    count = _frame_4k_count(n) + 13
    cur = torch.randint(0, 256, (count, n * n), dtype=torch.uint8, device=dev, generator=g)
    noise = torch.randint(-30, 31, (count, n * n), dtype=torch.int16, device=dev, generator=g)
    ref = (cur.to(torch.int16) + noise).clamp_(0, 255).to(torch.uint8)
    ref[::11] = 255 - cur[::11]                                 # large differences as well
    out = [torch.empty(count, dtype=torch.int32, device=dev) for _ in range(2)]
    _lib.check(L.kvz_hip_sad_nxn_batch(n, cur.data_ptr(), ref.data_ptr(), count, out[0].data_ptr(), None), "sad")
    _lib.check(L.kvz_hip_satd_nxn_batch(n, cur.data_ptr(), ref.data_ptr(), count, out[1].data_ptr(), None), "satd")
    _lib.check(L.kvz_hip_stream_sync(None), "sync")
    a, b = cur.cpu().numpy(), ref.cpu().numpy()
    np.testing.assert_array_equal(out[0].cpu().numpy().view(np.uint32), O.cost_nxn_many("sad", n, a, b))
    np.testing.assert_array_equal(out[1].cpu().numpy().view(np.uint32), O.cost_nxn_many("satd", n, a, b))
    # dual entries (sad / satd_NxN_dual): two predictions per original, contiguous here (pred_stride n*n, item stride 2 n*n)
    items = count // 2
    dual = [torch.empty(2 * items, dtype=torch.int32, device=dev) for _ in range(2)]
    orig = ref[:items].contiguous()
    _lib.check(L.kvz_hip_sad_nxn_dual_batch(n, cur.data_ptr(), n * n, 2 * n * n, orig.data_ptr(), items, dual[0].data_ptr(), None), "sad dual")
    _lib.check(L.kvz_hip_satd_nxn_dual_batch(n, cur.data_ptr(), n * n, 2 * n * n, orig.data_ptr(), items, dual[1].data_ptr(), None), "satd dual")
    _lib.check(L.kvz_hip_stream_sync(None), "sync")
    o2 = np.repeat(orig.cpu().numpy(), 2, axis=0)
    np.testing.assert_array_equal(dual[0].cpu().numpy().view(np.uint32), O.cost_nxn_many("sad", n, a[:2 * items], o2))
    np.testing.assert_array_equal(dual[1].cpu().numpy().view(np.uint32), O.cost_nxn_many("satd", n, a[:2 * items], o2))


@pytest.mark.parametrize("kind,n", [("dct", 4), ("dct", 8), ("dct", 16), ("dct", 32), ("idct", 4), ("idct", 8), ("idct", 16), ("idct", 32),
                                    ("dst", 4), ("idst", 4)])
def test_transforms_4k_frame_every_size(env, kind, n):
    """one 3840x2160 frame's worth of n x n residual / coefficient blocks (+ 13), incl. +-32768 extremes, every output exact"""
    torch, _lib, L, dev, g = env
    count = _frame_4k_count(n) + 13
    x = torch.randint(-255, 256, (count, n * n), dtype=torch.int16, device=dev, generator=g)
    x[::9] = torch.randint(-32768, 32768, x[::9].shape, dtype=torch.int32, device=dev, generator=g).to(torch.int16)
    x[3::101] = 32767; x[5::101] = -32768
    y = torch.empty_like(x)
    _lib.check(L.kvz_hip_transform_batch({"dct": 0, "idct": 1, "dst": 2, "idst": 3}[kind], n, x.data_ptr(), y.data_ptr(), count, None), kind)
    _lib.check(L.kvz_hip_stream_sync(None), "sync")
    np.testing.assert_array_equal(y.cpu().numpy(), O.transform_many(kind, n, x.cpu().numpy()))


@pytest.mark.parametrize("width", [4, 8, 16, 32])
def test_quantize_residual_frame_is_consistent_with_the_separate_entries(env, width):
    """every TU of 8 1080p frames through the fused entry; the same result must come out of the chain of separate entries
    (residual -> dct -> quant -> dequant -> idct -> reconstruct), TUs without coefficients must keep their prediction, and
    every TU must equal the oracle"""
    torch, _lib, L, dev, g = env
    from kvazaar_amd._lib import QuantParams
    n = (1920 // width) * (1080 // width) * 8
    px = n * width * width
    ref = torch.randint(0, 256, (px,), dtype=torch.uint8, device=dev, generator=g)
    noise = torch.randint(-20, 21, (px,), dtype=torch.int16, device=dev, generator=g)
    noise.view(n, -1)[::3] //= 8                              # a third of the TUs almost predicted: many all-zero TUs
    pred = (ref.to(torch.int16) + noise).clamp_(0, 255).to(torch.uint8)
    qp = QuantParams(); qp.qp = 32
    rec = torch.empty_like(ref); coef = torch.empty(px, dtype=torch.int16, device=dev); has = torch.empty(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()                                  # the library's stream does not wait for torch's: inputs must be complete
    _lib.check(L.kvz_hip_quantize_residual_batch(C.byref(qp), 0, width, 0, 0, 0, ref.data_ptr(), pred.data_ptr(), rec.data_ptr(),
                                                 coef.data_ptr(), has.data_ptr(), n, None), "quantize_residual")
    res = torch.empty(px, dtype=torch.int16, device=dev); t1 = torch.empty_like(res); q = torch.empty_like(res)
    dq = torch.empty_like(res); back = torch.empty_like(res); rec2 = torch.empty_like(ref)
    _lib.check(L.kvz_hip_residual_batch(ref.data_ptr(), pred.data_ptr(), res.data_ptr(), px, None), "residual")
    _lib.check(L.kvz_hip_transform_batch(0, width, res.data_ptr(), t1.data_ptr(), n, None), "dct")
    _lib.check(L.kvz_hip_quant_batch(C.byref(qp), t1.data_ptr(), q.data_ptr(), width, 0, 0, n, None), "quant")
    _lib.check(L.kvz_hip_dequant_batch(C.byref(qp), q.data_ptr(), dq.data_ptr(), width, 0, n, None), "dequant")
    _lib.check(L.kvz_hip_transform_batch(1, width, dq.data_ptr(), back.data_ptr(), n, None), "idct")
    _lib.check(L.kvz_hip_reconstruct_batch(back.data_ptr(), pred.data_ptr(), rec2.data_ptr(), px, None), "reconstruct")
    _lib.check(L.kvz_hip_stream_sync(None), "sync")
    assert bool((coef == q).all())
    nz = q.view(n, -1).ne(0).any(dim=1)
    assert bool((has.ne(0) == nz).all()) and 0 < int(nz.sum()) < n
    # with coefficients: the chain's reconstruction; without: the prediction (quant-generic.c:262-271)
    want = torch.where(nz[:, None], rec2.view(n, -1), pred.view(n, -1))
    assert bool((rec.view(n, -1) == want).all())
    w2 = width * width
    r, c, h = O.quantize_residual_many(ref.view(n, w2).cpu().numpy(), pred.view(n, w2).cpu().numpy(), width, 32, 0, 0, 0)
    np.testing.assert_array_equal(rec.view(n, w2).cpu().numpy(), r)
    np.testing.assert_array_equal(coef.view(n, w2).cpu().numpy(), c)
    np.testing.assert_array_equal(has.cpu().numpy() != 0, h != 0)


def test_sample_luma_integer_position_is_the_identity_over_a_frame(env):
    """fractional offset (0, 0) runs both filter passes with the taps {0,0,0,64,...}: every 8x8 and 16x16 block of a 1080p
    frame -- incl. windows clamped at the frame border -- must come back unchanged, 8-bit and (<< 6) 14-bit"""
    torch, _lib, L, dev, g = env
    W, H = 1920, 1080
    frame = torch.randint(0, 256, (H, W), dtype=torch.uint8, device=dev, generator=g)
    for n in (8, 16):
        blocks = np.array([(x, y, 0, 0, n, n) for y in range(0, H - n + 1, n) for x in range(0, W, n)], dtype=np.int32)
        bd = torch.from_numpy(blocks).to(dev)
        offs = torch.arange(len(blocks), dtype=torch.int64, device=dev) * (n * n)
        out8 = torch.empty(len(blocks) * n * n, dtype=torch.uint8, device=dev)
        out14 = torch.empty(len(blocks) * n * n, dtype=torch.int16, device=dev)
        torch.cuda.synchronize()                              # the library's stream does not wait for torch's: inputs must be complete
        _lib.check(L.kvz_hip_sample_luma_batch(frame.data_ptr(), W, W, H, bd.data_ptr(), offs.data_ptr(), len(blocks), 0, out8.data_ptr(), None), "sample")
        _lib.check(L.kvz_hip_sample_luma_batch(frame.data_ptr(), W, W, H, bd.data_ptr(), offs.data_ptr(), len(blocks), 1, out14.data_ptr(), None), "sample14")
        _lib.check(L.kvz_hip_stream_sync(None), "sync")
        rows = (H // n) * n
        want = frame[:rows].view(rows // n, n, W // n, n).permute(0, 2, 1, 3).reshape(-1)
        assert bool((out8 == want).all())
        assert bool((out14 == want.to(torch.int16) * 64).all())


def test_search_pu_results_do_not_depend_on_the_batch(env):
    """every 16x16 PU of a 1080p frame in one launch, and the same PUs in three launches of shuffled thirds: a PU's result
    depends on nothing but its descriptor"""
    torch, _lib, L, dev, g = env
    from patterns import me_frames, me_params
    W, H = 1920, 1080
    pic_np, ref_np = me_frames(W, H, 99, (3, -2))
    pic, ref = torch.from_numpy(pic_np).to(dev), torch.from_numpy(ref_np).to(dev)
    xy = [(x, y) for y in range(0, H - 15, 16) for x in range(0, W, 16)]
    pus = np.zeros((len(xy), 16), dtype=np.int32)
    pus[:, 0] = [p[0] for p in xy]; pus[:, 1] = [p[1] for p in xy]; pus[:, 2] = 16; pus[:, 3] = 16
    prm = me_params()
    pd = torch.from_numpy(pus).to(dev)
    whole = torch.empty((len(xy), 8), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()                                  # the library's stream does not wait for torch's: inputs must be complete
    _lib.check(L.kvz_hip_search_pu_batch(pic.data_ptr(), W, W, H, ref.data_ptr(), W, W, H, pd.data_ptr(), len(xy), prm.ctypes.data,
                                         whole.data_ptr(), None), "search_pu")
    perm = torch.randperm(len(xy), device=dev, generator=g)
    parts = torch.empty_like(whole)
    for part in perm.chunk(3):
        sub = pd[part].contiguous()
        out = torch.empty((len(part), 8), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        _lib.check(L.kvz_hip_search_pu_batch(pic.data_ptr(), W, W, H, ref.data_ptr(), W, W, H, sub.data_ptr(), len(part), prm.ctypes.data,
                                             out.data_ptr(), None), "search_pu part")
        _lib.check(L.kvz_hip_stream_sync(None), "sync")
        parts[part] = out
    _lib.check(L.kvz_hip_stream_sync(None), "sync")
    assert bool((whole == parts).all())
    assert int((whole[:, 2] != -1).sum()) == len(xy)          # every PU was searched (cost field set)


def test_sao_edge_statistics_add_up_over_a_frame(env):
    """every 64x64 luma LCU of a 1080p frame: per class the five category counts add up to the interior pixel count and
    the five sums to the interior's sum of (orig - rec), both computed independently with torch; a sample equals the oracle"""
    torch, _lib, L, dev, g = env
    n = 30 * 16
    orig = torch.randint(0, 256, (n, 64, 64), dtype=torch.uint8, device=dev, generator=g)
    rec = (orig.to(torch.int16) + torch.randint(-6, 7, orig.shape, dtype=torch.int16, device=dev, generator=g)).clamp_(0, 255).to(torch.uint8)
    stats = torch.empty((n, 4, 2, 5), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()                                  # the library's stream does not wait for torch's: inputs must be complete
    _lib.check(L.kvz_hip_sao_edge_stats_batch(orig.data_ptr(), rec.data_ptr(), 64, 64, n, stats.data_ptr(), None), "sao_edge_stats")
    _lib.check(L.kvz_hip_stream_sync(None), "sync")
    diff = (orig.to(torch.int32) - rec.to(torch.int32))[:, 1:-1, 1:-1].sum(dim=(1, 2))
    assert bool((stats[:, :, 1, :].sum(dim=2) == 62 * 62).all())
    assert bool((stats[:, :, 0, :].sum(dim=2) == diff[:, None]).all())
    for i in (0, 7, n - 1):
        for eo in range(4):
            np.testing.assert_array_equal(stats[i, eo].cpu().numpy(),
                                          O.calc_sao_edge_dir(orig[i].cpu().numpy().ravel(), rec[i].cpu().numpy().ravel(), eo, 64, 64))


@pytest.mark.parametrize("slice_is_b", [0, 1])
def test_deblock_a_1080p_frame(env, slice_is_b):
    """a whole 1920 x 1088 4:2:0 frame with a random CU / PU / TU quadtree against the oracle"""
    torch, _lib, L, dev, g = env
    from kvazaar_amd import api
    from patterns import deblock_case, deblock_params
    prm = deblock_params(qp=35, per_cu_qp=1, slice_is_b=slice_is_b, beta=1, tc=-1)
    y, u, v, cus = deblock_case(1920, 1088, 2024 + slice_is_b, slice_is_b=slice_is_b, qp=35)
    want = O.deblock_frame(y, u, v, cus, prm)
    torch.cuda.synchronize()                                  # the library's stream does not wait for torch's: inputs must be complete
    got = api.deblock_frame(y, u, v, cus, prm)
    for a, b, name in zip(got, want, "yuv"):
        np.testing.assert_array_equal(a, b, err_msg=name)
    assert int((want[0] != y).sum()) > 100000
