"""Stream-K of fino_gemm_ws (csrc/fino_gemm.hip: gemm_sk_kernel): the K-tile units of the last one-to-two rounds' tiles
-- or of every tile when the GEMM has fewer tiles than CUs, the token-shard shapes -- dealt to all CUs inside one launch;
a split tile's pieces meet in the block that holds its first K-tiles through a write-through fp32 hand-off.

Checked here: every output element against the fp32 reference and against the whole-tile path (only the fp32 summation
order of split tiles may differ), at the shard shapes (M = 1540 / 3080 / 6160 rows), at the one-GPU shapes (12320 /
24640), for every epilogue; determinism over repeated launches; the hand-off under UNEVEN load (two streams running
stream-K GEMMs on their own workspaces beside a third stream of unrelated kernels), every word compared; the flag area
left zero by every launch; and a captured hipGraph replay."""
import pytest
import torch

DEV = "cuda"
from tests.parity import rel_rms
from tests.test_kernels_gpu import gemm_ref

pytestmark = pytest.mark.gpu
D, FF = 3072, 14336


@pytest.fixture(scope="module")
def lib():
    from frameino_amd import _lib
    return _lib.lib()


def _operands(m, n, k, epi, seed=0, dtype=torch.bfloat16):
    g = torch.Generator(device=DEV).manual_seed(seed)
    a = torch.randn(m, k, device=DEV, generator=g).to(dtype)
    w = (torch.randn(n, k, device=DEV, generator=g) * k ** -0.5).to(dtype)
    bias = torch.randn(n, device=DEV, generator=g).to(dtype)
    res = torch.randn(m, n, device=DEV, generator=g).to(dtype) if epi >= 2 else None
    gate = torch.randn(2, n, device=DEV, generator=g) if epi >= 3 else None
    sel = (torch.arange(m, device=DEV) % 5 == 0).to(torch.int32) if epi >= 3 else None
    return a, w, bias, res, gate, sel


def _both(lib, m, n, k, epi, dtype=torch.bfloat16, seed=0):
    from frameino_amd import ops
    a, w, bias, res, gate, sel = _operands(m, n, k, epi, seed, dtype)
    try:
        lib.fino_tune_set(3, 2)                              # stream-K whenever legal
        assert lib.fino_gemm_workspace_bytes(m, n, k) > 0, "shape has no partial round: pick another"
        sk = ops.gemm(a, w, bias, epi, res, gate, sel)
        lib.fino_tune_set(3, 1)                              # whole tiles only
        assert lib.fino_gemm_workspace_bytes(m, n, k) == 0
        whole = ops.gemm(a, w, bias, epi, res, gate, sel)
    finally:
        lib.fino_tune_set(3, 0)
    return sk, whole, (a, w, bias, res, gate, sel)


@pytest.mark.parametrize("m", [1540, 3080, 6160], ids=lambda m: f"shard{m}")
@pytest.mark.parametrize("n,k,epi", [(D, D, 3), (D, D, 0), (3 * D, D, 0), (2 * D, D, 0), (FF, D, 1), (D, FF, 3), (D, D, 2)],
                         ids=["out", "q", "qkv", "kv", "ffn_up", "ffn_down", "out2"])
def test_stream_k_at_the_token_shard_shapes(lib, m, n, k, epi):
    """every block GEMM of a Wan2.2-5B layer at the 8- / 4- / 2-way token-shard row counts"""
    tiles = -(-m // 256) * -(-n // 256)
    if tiles % 256 == 0:
        pytest.skip("whole number of rounds")
    sk, whole, ops_ = _both(lib, m, n, k, epi)
    ref = gemm_ref(ops_[0], ops_[1], ops_[2], epi, ops_[3], ops_[4], ops_[5])
    assert torch.isfinite(sk.float()).all()
    r_sk, r_whole = rel_rms(sk, ref.float()), rel_rms(whole, ref.float())
    assert r_sk < 2.0 ** -7 and r_sk < 1.2 * r_whole + 1e-5, (r_sk, r_whole)
    assert rel_rms(sk, whole.float()) < 2.0 ** -9


@pytest.mark.parametrize("m,n,k,epi", [(24640, D, D, 3), (24640, D, FF, 3), (12320, D, FF, 3), (12320, 3 * D, D, 0),
                                       (24640, FF, D, 1)], ids=["out_B2", "ffn_down_B2", "ffn_down_B1", "qkv_B1", "ffn_up_B2"])
def test_stream_k_at_the_one_gpu_shapes(lib, m, n, k, epi):
    sk, whole, ops_ = _both(lib, m, n, k, epi)
    rows = torch.tensor(sorted({0, 255, 256, m - 1, m - 33, m - 257} | set(torch.randint(0, m, (96,)).tolist())), device=DEV)
    a, w, bias, res, gate, sel = ops_
    ref = gemm_ref(a[rows], w, bias, epi, None if res is None else res[rows], gate, None if sel is None else sel[rows])
    assert rel_rms(sk[rows], ref.float()) < 2.0 ** -7
    assert rel_rms(sk, whole.float()) < 2.0 ** -9
    assert not torch.equal(sk, whole)          # it did take the other path (some tile was summed in two pieces)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,n,k", [(700, 512, 1024), (300, 256, 4096), (257, 264, 512), (2000, 768, 576), (513, 1024, 8192)])
def test_stream_k_small_and_ragged(lib, dtype, m, n, k):
    """few tiles (all of them streamed), ragged M / N edges, short K (pieces of a handful of K-tiles, snapped ranges), long K
    (a tile cut into many pieces: its owner collects several slots)"""
    for epi in (0, 3):
        sk, whole, ops_ = _both(lib, m, n, k, epi, dtype)
        ref = gemm_ref(ops_[0], ops_[1], ops_[2], epi, ops_[3], ops_[4], ops_[5])
        assert rel_rms(sk, ref.float()) < (2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10)
        assert rel_rms(sk, whole.float()) < 2.0 ** -9


def test_stream_k_is_deterministic_and_leaves_its_flags_zero(lib):
    from frameino_amd import ops
    m, n, k, epi = 3080, D, D, 3
    a, w, bias, res, gate, sel = _operands(m, n, k, epi, seed=3)
    try:
        lib.fino_tune_set(3, 2)
        first = ops.gemm(a, w, bias, epi, res, gate, sel)
        for _ in range(8):
            assert torch.equal(ops.gemm(a, w, bias, epi, res, gate, sel), first)
        torch.cuda.synchronize()
        key = (a.device.index, torch.cuda.current_stream().cuda_stream)
        ws = ops._gemm_ws[key]
        assert int(ws[:64 * 1024 // 4].view(torch.int32).abs().sum()) == 0
    finally:
        lib.fino_tune_set(3, 0)


def test_stream_k_hand_off_under_uneven_load_every_word(lib):
    """two streams run stream-K GEMMs concurrently (own workspaces), a third keeps unrelated CUs busy for uneven spans;
    every launch's output equals, word for word, what the same GEMM gives alone"""
    from frameino_amd import ops
    cases = [(3080, D, D, 3), (1540, D, FF, 3), (3080, 2 * D, D, 0)]
    data = [_operands(m, n, k, e, seed=10 + i) for i, (m, n, k, e) in enumerate(cases)]
    noise = torch.randn(4096, 4096, device=DEV)
    try:
        lib.fino_tune_set(3, 2)
        alone = [ops.gemm(d[0], d[1], d[2], c[3], d[3], d[4], d[5]) for c, d in zip(cases, data)]
        torch.cuda.synchronize()
        s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
        outs = []
        for rep in range(6):
            with torch.cuda.stream(s3):
                for _ in range(1 + rep % 3):
                    noise = (noise @ noise).clamp_(-1, 1)
            for st_, order in ((s1, (0, 1, 2)), (s2, (2, 0, 1))):
                with torch.cuda.stream(st_):
                    for i in order:
                        c, d = cases[i], data[i]
                        outs.append((i, ops.gemm(d[0], d[1], d[2], c[3], d[3], d[4], d[5])))
        torch.cuda.synchronize()
    finally:
        lib.fino_tune_set(3, 0)
    for i, o in outs:
        assert torch.equal(o, alone[i]), f"case {cases[i]} differs under load"


def test_stream_k_replays_from_a_hip_graph(lib):
    from frameino_amd import ops
    m, n, k, epi = 1540, D, D, 3
    a, w, bias, res, gate, sel = _operands(m, n, k, epi, seed=5)
    out = torch.empty(m, n, device=DEV, dtype=torch.bfloat16)
    try:
        lib.fino_tune_set(3, 2)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            eager = ops.gemm(a, w, bias, epi, res, gate, sel).clone()
            ops.gemm(a, w, bias, epi, res, gate, sel, out=out)          # warm: workspace of this stream exists
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                ops.gemm(a, w, bias, epi, res, gate, sel, out=out)
            for _ in range(3):
                out.zero_()
                g.replay()
                torch.cuda.synchronize()
                assert torch.equal(out, eager)
    finally:
        lib.fino_tune_set(3, 0)


def test_default_policy_splits_the_shard_shapes_and_not_a_full_round(lib):
    assert lib.fino_gemm_workspace_bytes(3080, D, D) > 0           # 156 tiles on 256 CUs
    assert lib.fino_gemm_workspace_bytes(24640, D, FF) > 0         # 4.55 rounds of long tiles
    assert lib.fino_gemm_workspace_bytes(1540, 3 * D, D) == 0      # 252 tiles: one full round
    assert lib.fino_gemm_workspace_bytes(256 * 16, 256 * 16, D) == 0
