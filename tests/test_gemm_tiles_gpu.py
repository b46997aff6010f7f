"""Tile heights of fino_gemm (csrc/fino_gemm.hip: gemm_pp_kernel<..., MI>, plan_tiles): the leading rows run as 256 x 256
tiles in whole rounds of the CUs, the rest as ONE more launch of 64 .. 256-row tiles that fits one more round -- 3080 rows
(a 4-way token shard) are 240 tiles of 160 rows instead of 156 of 256 on 256 CUs.

Every output element is one fp32 dot product over K in the same order whatever the tile height, so every tiling must give
BIT-IDENTICAL results: checked for every forced height against the 256-row path, on all elements, for every epilogue,
ragged M / N, the token-shard and one-GPU row counts; plus the fp32 reference, the planner's choices, and the two-launch
form (rows_256 > 0 and a remainder) through strided views and an in-place residual."""
import pytest
import torch

from tests.parity import rel_rms
from tests.test_kernels_gpu import gemm_ref

pytestmark = pytest.mark.gpu
DEV = "cuda"
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


def _run(lib, ops_, epi, force):
    from frameino_amd import ops
    a, w, bias, res, gate, sel = ops_
    try:
        lib.fino_tune_set(3, force)
        return ops.gemm(a, w, bias, epi, res, gate, sel)
    finally:
        lib.fino_tune_set(3, 0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,n,k", [(700, 512, 1024), (300, 256, 192), (257, 264, 512), (2000, 768, 576), (63, 3072, 128),
                                   (1, 256, 64), (1540, 3072, 3072)])
def test_every_tile_height_is_bit_identical_to_the_256_row_path(lib, dtype, m, n, k):
    for epi in (0, 1, 2, 3, 4):
        ops_ = _operands(m, n, k, epi, seed=epi, dtype=dtype)
        base = _run(lib, ops_, epi, 8)
        ref = gemm_ref(*ops_[:3], epi, *ops_[3:])
        assert rel_rms(base, ref.float()) < (2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10)
        for mi in (2, 3, 4, 5, 6, 7):
            out = _run(lib, ops_, epi, mi)
            assert torch.equal(out, base), f"tile height {32 * mi}, epilogue {epi}: differs from the 256-row tiles"
        assert torch.equal(_run(lib, ops_, epi, 0), base)          # and so does whatever the planner picks


@pytest.mark.parametrize("m", [1540, 3080, 6160, 12320, 24640], ids=lambda m: f"rows{m}")
@pytest.mark.parametrize("n,k,epi", [(D, D, 3), (3 * D, D, 0), (FF, D, 1), (D, FF, 3)], ids=["out", "qkv", "ffn_up", "ffn_down"])
def test_planned_tiling_at_the_model_shapes(lib, m, n, k, epi):
    """the block GEMMs of a Wan2.2-5B layer at the one-GPU row counts and the 2- / 4- / 8-way token-shard ones: the planned
    tiling (possibly two launches) == 256-row tiles, bit for bit, and sampled rows against fp32"""
    ops_ = _operands(m, n, k, epi, seed=1)
    planned, base = _run(lib, ops_, epi, 0), _run(lib, ops_, epi, 8)
    assert torch.equal(planned, base)
    rows = torch.tensor(sorted({0, 255, 256, m - 1, m - 33, m - 257} | set(torch.randint(0, m, (64,)).tolist())), device=DEV)
    a, w, bias, res, gate, sel = ops_
    ref = gemm_ref(a[rows], w, bias, epi, None if res is None else res[rows], gate, None if sel is None else sel[rows])
    assert rel_rms(planned[rows], ref.float()) < 2.0 ** -7


def test_the_planner(lib):
    from frameino_amd import ops
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    # a 4-way token shard on 12 tile columns: 156 tiles of 256 rows would leave 100 CUs idle -> lower tiles, one launch
    r256, rest = ops.gemm_plan(3080, D)
    assert r256 == 0 and 64 <= rest < 256 and -(-3080 // rest) * 12 <= cus
    # an 8-way shard: 84 tiles -> lower tiles
    r256, rest = ops.gemm_plan(1540, D)
    assert r256 == 0 and 64 <= rest < 256
    # a whole number of rounds stays as it is
    assert ops.gemm_plan(256 * 16, 256 * 16) == (256 * 16, 0)
    # the plan covers exactly the rows there are, in legal heights
    for m in (1, 255, 256, 257, 1540, 12320, 24640, 19126 * 2):
        for n in (192, 3072, 9216, 14336):
            r256, rest = ops.gemm_plan(m, n)
            assert 0 <= r256 <= m and (r256 % 256 == 0 or r256 == m)
            assert (rest == 0) == (r256 == m) and rest % 32 == 0 and rest <= 256


def test_two_launch_form_on_strided_views_with_an_in_place_residual(lib):
    """rows_256 > 0 and a remainder: the second launch starts at a row offset of A, C, R and the selector"""
    from frameino_amd import ops
    m, d = 24640 // 4 + 70, 512
    r256, rest = ops.gemm_plan(m, d)
    g = torch.Generator(device=DEV).manual_seed(4)
    big = torch.randn(m, 3 * d, device=DEV, generator=g).bfloat16()
    w = (torch.randn(d, d, device=DEV, generator=g) * 0.05).bfloat16()
    x = torch.randn(m, d, device=DEV, generator=g).bfloat16()
    gate = torch.randn(2, d, device=DEV, generator=g)
    sel = (torch.arange(m, device=DEV) % 3 == 0).to(torch.int32)
    ref = gemm_ref(big[:, d:2 * d], w, None, 3, x.clone(), gate, sel)
    ops.gemm(big[:, d:2 * d], w, None, 3, residual=x, gate=gate, sel=sel, out=x)      # C aliases R, A is a column slice
    assert rel_rms(x, ref.float()) < 2.0 ** -7
    print(f"plan for {m} x {d}: {r256} rows of 256-row tiles + {rest}-row tiles")


@pytest.mark.parametrize("m", [1540, 3080, 700])
def test_split_n_output_equals_the_column_slices_of_one_output(lib, m):
    """fino_gemm_split_n (the fused q | k | v projection of a token shard: q to its buffer, k | v to the gather's send
    buffer, row-strided destinations): bit-equal to the slices of the ordinary GEMM"""
    from frameino_amd import ops
    d = 512
    a, w, bias, _, _, _ = _operands(m, 3 * d, 1024, 0, seed=9)
    whole = ops.gemm(a, w, bias)
    q = torch.zeros(m + 3, d, device=DEV, dtype=torch.bfloat16)
    kv = torch.zeros(m + 5, 2 * d + 64, device=DEV, dtype=torch.bfloat16)          # padded rows: a strided destination
    ops.gemm(a, w, bias, out=q[:m], out2=kv[:m, :2 * d], split=d)
    assert torch.equal(q[:m], whole[:, :d]) and torch.equal(kv[:m, :2 * d], whole[:, d:])
    assert float(q[m:].abs().max()) == 0.0 and float(kv[m:].abs().max()) == 0.0 and float(kv[:, 2 * d:].abs().max()) == 0.0
    with pytest.raises(RuntimeError, match="n_split"):
        ops.gemm(a, w, bias, out=q[:m], out2=kv[:m, :2 * d], split=d + 8)


@pytest.mark.parametrize("ways,n,lpad", [(4, 3080, 3080), (8, 1540, 1540), (2, 6160, 6160), (4, 1000, 1024)])
def test_blocked_a_out_projection_equals_the_permute_copy_plus_gemm(ways, n, lpad):
    """fino_gemm_blocked_a: the gated-residual out-projection reading [peer][token][heads of that peer] (what the heads
    all-to-all returns) == the permute copy into [token, D] followed by fino_gemm, bit for bit"""
    from frameino_amd import ops
    d = 3072
    dp = d // ways
    g = torch.Generator(device=DEV).manual_seed(ways * 1000 + n)
    orv = torch.randn(ways, lpad, dp, device=DEV, generator=g).bfloat16()
    w = (torch.randn(d, d, device=DEV, generator=g) * 0.02).bfloat16()
    b = torch.randn(d, device=DEV, generator=g).bfloat16()
    x = torch.randn(n, d, device=DEV, generator=g).bfloat16()
    gate = torch.randn(2, d, device=DEV, generator=g)
    sel = (torch.arange(n, device=DEV) % 2).to(torch.int32)
    att = orv[:, :n].permute(1, 0, 2).reshape(n, d).contiguous()
    want = ops.gemm(att, w, b, ops.EPI_GATED_RESIDUAL, x, gate, sel)
    got = ops.gemm_blocked_a(orv, n, w, b, x, gate, sel, out=torch.empty_like(x))
    assert torch.equal(got, want)
    x2 = x.clone()
    ops.gemm_blocked_a(orv, n, w, b, x2, gate, sel, out=x2)          # in place on the residual, as the forward calls it
    assert torch.equal(x2, want)
    # two head groups: K block j * 2 + g comes from the return buffer of group g, slice j
    if (dp // 128) % 2 == 0:
        dg = dp // 2
        flat = torch.empty(2, ways, lpad, dg, device=DEV, dtype=torch.bfloat16)
        flat[0].copy_(orv[:, :, :dg])
        flat[1].copy_(orv[:, :, dg:])
        got2 = ops.gemm_blocked_a(flat, n, w, b, x, gate, sel, out=torch.empty_like(x))
        assert torch.equal(got2, want)


@pytest.mark.parametrize("blocks,blk_k", [(16, 192), (5, 576), (3, 1088), (7, 448)])
def test_blocked_a_with_non_power_of_two_blocks(blocks, blk_k):
    """the K-tile -> block index of the K-blocked A is a reciprocal multiply ((kt * ceil(65536 / a_tpb)) >> 16);
    a_tpb = 3, 9, 17, 7 K-tiles per block (never a power of two) and odd block counts, against the permute copy + fino_gemm"""
    from frameino_amd import ops
    n, rows, lpad = 512, 700, 704
    k = blocks * blk_k
    g = torch.Generator(device=DEV).manual_seed(blocks * 7 + blk_k)
    orv = torch.randn(blocks, lpad, blk_k, device=DEV, generator=g).bfloat16()
    w = (torch.randn(n, k, device=DEV, generator=g) * 0.03).bfloat16()
    b = torch.randn(n, device=DEV, generator=g).bfloat16()
    x = torch.randn(rows, n, device=DEV, generator=g).bfloat16()
    gate = torch.randn(2, n, device=DEV, generator=g)
    sel = (torch.arange(rows, device=DEV) % 2).to(torch.int32)
    a = orv[:, :rows].permute(1, 0, 2).reshape(rows, k).contiguous()
    want = ops.gemm(a, w, b, ops.EPI_GATED_RESIDUAL, x, gate, sel)
    got = ops.gemm_blocked_a(orv, rows, w, b, x, gate, sel, out=torch.empty_like(x))
    assert torch.equal(got, want)
    for tm in (8, 3):                                         # the per-call tile height does not change a bit either
        assert torch.equal(ops.gemm_blocked_a(orv, rows, w, b, x, gate, sel, out=torch.empty_like(x), tile_m=tm), want)
        assert torch.equal(ops.gemm(a, w, b, ops.EPI_GATED_RESIDUAL, x, gate, sel, tile_m=tm), want)


@pytest.mark.parametrize("m", [8, 65, 129, 300, 3080, 1540])
@pytest.mark.parametrize("epi", ["bias", "gelu", "gated"])
def test_ragged_last_tile_row_runs_as_a_lower_tile_bit_identically(m, epi):
    """Round 6 (`GP_RAGGED`): a launch of 256-row tiles runs a last tile row of <= 64 / <= 128 valid rows as the 64- / 128-row
    instantiation of the same loop and epilogue.  Bit-equal to the one-height launches of 128-row tiles and to the planned launch, and
    to a torch fp32 reference within bf16 rounding."""
    from frameino_amd import ops
    g = torch.Generator(device=DEV).manual_seed(m)
    n, k = 768, 1024
    a = torch.randn(m, k, device=DEV, generator=g).bfloat16()
    w = (torch.randn(n, k, device=DEV, generator=g) * 0.05).bfloat16()
    b = torch.randn(n, device=DEV, generator=g).bfloat16()
    res = torch.randn(m, n, device=DEV, generator=g).bfloat16()
    gate = torch.randn(2, n, device=DEV, generator=g)
    sel = (torch.arange(m, device=DEV) % 2).to(torch.int32)
    kw = {"bias": dict(epilogue=ops.EPI_NONE), "gelu": dict(epilogue=ops.EPI_GELU_TANH),
          "gated": dict(epilogue=ops.EPI_GATED_RESIDUAL, residual=res, gate=gate, sel=sel)}[epi]
    outs = [ops.gemm(a, w, b, tile_m=t, **kw) for t in (8, 4, 0)]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    lin = torch.nn.functional.linear(a.float(), w.float(), b.float())
    ref = {"bias": lin, "gelu": torch.nn.functional.gelu(lin.bfloat16().float(), approximate="tanh"),
           "gated": res.float() + lin.bfloat16().float() * gate[sel.long()]}[epi]
    assert rel_rms(outs[0], ref) < 4e-3
