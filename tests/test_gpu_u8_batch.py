"""Multi-query MFMA path (qamd_u8_*_batch): every score / neighbour list must be bit-identical
to the single-query path, which is itself pinned to the oracle (test_gpu_u8.py)."""
import numpy as np
import pytest

from util import assert_bits_equal

pytestmark = pytest.mark.gpu

qa = pytest.importorskip("quantization_amd")
D = qa.DistanceType


@pytest.mark.parametrize("n,dim,nq", [(1000, 768, 5), (300, 65, 130), (5000, 1536, 64), (129, 16, 3),
                                      (700, 100, 257), (64, 2048, 2), (2500, 128, 1),
                                      # ping-pong kernel, 256-query tile: partial row tiles, two query tiles
                                      (600, 200, 300), (100, 256, 130), (1025, 144, 513),
                                      # small batches (padded to one 128-query MFMA tile)
                                      (3000, 256, 9), (500, 1024, 16), (800, 1536, 7), (400, 512, 12),
                                      (2000, 768, 16), (777, 768, 1), (900, 2048, 5)])
@pytest.mark.parametrize("dist,invert", [(D.Dot, False), (D.L2, False), (D.Dot, True)])
def test_score_batch_equals_single_query_and_oracle(qo, n, dim, nq, dist, invert):
    rng = np.random.default_rng(n + dim + nq)
    data = rng.random((n, dim), dtype=np.float32)
    queries = rng.random((nq, dim), dtype=np.float32)
    enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, dist, invert))
    batch = enc.encode_query_batch(queries)
    got = enc.score_batch(batch)
    assert got.shape == (nq, n)
    rows, meta = qo.u8_encode(data, int(dist), invert)
    for qi in sorted({0, nq // 2, nq - 1}):
        codes, qoff = qo.u8_encode_query(meta, queries[qi])
        want = qo.u8_score_all(meta, rows, codes, qoff, order=qo.ORDER_SIMPLE)
        assert_bits_equal(got[qi], want, f"query {qi} vs oracle")
    single = np.stack([enc.score_all(enc.encode_query(queries[qi])) for qi in range(min(nq, 8))])
    assert_bits_equal(got[: single.shape[0]], single, "batch vs single-query path")


def test_batch_l1_runs_the_single_query_kernel():
    """sum |q - v| is not a contraction (no MFMA form): the batch API serves L1 by looping the
    single-query scan / top-k, so the results are those, bit for bit."""
    rng = np.random.default_rng(3)
    n, dim, nq, k = 3000, 80, 5, 12
    data = rng.random((n, dim), dtype=np.float32)
    queries = rng.random((nq, dim), dtype=np.float32)
    enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, D.L1, False))
    b = enc.encode_query_batch(queries)
    got = enc.score_batch(b)
    ids, sc = enc.topk_batch(b, k, largest=False)
    for qi in range(nq):
        qobj = enc.encode_query(queries[qi])
        assert_bits_equal(got[qi], enc.score_all(qobj), f"L1 scores of query {qi}")
        want_ids, want_sc = enc.topk(qobj, k, largest=False)
        assert np.array_equal(ids[qi], want_ids) and np.array_equal(sc[qi].view(np.uint32), want_sc.view(np.uint32))


def test_device_queries_and_outputs(qo):
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(3)
    n, dim, nq = 4000, 256, 40
    data = rng.random((n, dim), dtype=np.float32)
    queries = rng.random((nq, dim), dtype=np.float32)
    enc = qa.EncodedVectorsU8.encode(torch.from_numpy(data).cuda(), qa.VectorParameters(dim, n, D.Dot, False))
    batch = enc.encode_query_batch(torch.from_numpy(queries).cuda())
    out = torch.empty((nq, n), dtype=torch.float32, device="cuda")
    enc.score_batch(batch, out=out)
    torch.cuda.synchronize()
    host = enc.score_batch(enc.encode_query_batch(queries))
    assert_bits_equal(out.cpu().numpy(), host, "device vs host queries")


@pytest.mark.parametrize("largest", [True, False])
def test_topk_batch_small_store_uses_exact_single_query_path(largest):
    rng = np.random.default_rng(5)
    n, dim, nq, k = 20000, 64, 9, 25
    data = rng.random((n, dim), dtype=np.float32)
    queries = rng.random((nq, dim), dtype=np.float32)
    enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, D.Dot, False))
    ids, sc = enc.topk_batch(enc.encode_query_batch(queries), k, largest=largest)
    for qi in range(nq):
        wi, ws = enc.topk(enc.encode_query(queries[qi]), k, largest=largest)
        assert np.array_equal(ids[qi], wi) and np.array_equal(sc[qi].view(np.uint32), ws.view(np.uint32))


@pytest.mark.parametrize("k,largest", [(30, True), (200, False)])
@pytest.mark.parametrize("dist,invert,lo", [(D.L2, False, 0.0), (D.Dot, False, 0.0), (D.Dot, True, 5.0), (D.L2, True, -3.0)])
@pytest.mark.parametrize("dim,nq", [(64, 150), (192, 300)])
def test_topk_batch_fused_large_store(k, largest, dist, invert, lo, dim, nq):
    """n >= 2^20: per-query pivots from the sampled sub-store, filter pass, per-query sort.
    dim 64 (rows of one K-tile) runs u8_gemm_kernel; dim 192 runs the ping-pong kernel: integer
    pre-filter folded into the accumulators in both directions (multiplier > 0 and < 0, largest
    and smallest), wave-private candidate lists, padding queries in the second query tile.
    `lo` shifts the data so that offset != 0 and vector_offset varies a lot inside a tile."""
    rng = np.random.default_rng(7)
    n = 1_300_000 if dim == 64 else 1_100_000
    data = rng.random((n, dim), dtype=np.float32) + np.float32(lo)
    queries = rng.random((nq, dim), dtype=np.float32) + np.float32(lo)
    enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, dist, invert))
    ids, sc = enc.topk_batch(enc.encode_query_batch(queries), k, largest=largest)
    for qi in (0, 1, 77, nq - 1):
        scores = enc.score_all(enc.encode_query(queries[qi]))
        order = np.lexsort((np.arange(n), -scores if largest else scores))[:k]
        assert np.array_equal(ids[qi], order.astype(np.uint32)), qi
        assert np.array_equal(sc[qi].view(np.uint32), scores[order].view(np.uint32))


@pytest.mark.parametrize("largest", [True, False])
@pytest.mark.parametrize("dist,invert", [(D.Dot, False), (D.L2, True)])
def test_topk_batch_prefilter_survives_huge_offsets(largest, dist, invert):
    """Data in [1000, 1001): alpha = 1/127 against |offset| = 1000, so scores are ~1e8 with an f32
    spacing of 8 while one code step moves a score by ~8: the f32 epilogue's rounding is as large
    as the quantity the integer pre-filter thresholds.  The pre-filter may only ever be too
    permissive: the batch result must still equal the exact single-query top-k bit for bit."""
    rng = np.random.default_rng(11)
    n, dim, nq, k = 1_100_000, 192, 140, 40
    data = rng.random((n, dim), dtype=np.float32) + np.float32(1000.0)
    queries = rng.random((nq, dim), dtype=np.float32) + np.float32(1000.0)
    enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, dist, invert))
    ids, sc = enc.topk_batch(enc.encode_query_batch(queries), k, largest=largest)
    for qi in (0, 3, nq - 1):
        want_ids, want_sc = enc.topk(enc.encode_query(queries[qi]), k, largest=largest)
        assert np.array_equal(ids[qi], want_ids), qi
        assert np.array_equal(sc[qi].view(np.uint32), want_sc.view(np.uint32)), qi


@pytest.mark.parametrize("nq,dim", [(6, 64), (6, 768), (13, 256)])
def test_topk_batch_small_batch_scan(nq, dim):
    """A handful of queries (padded to one 128-query MFMA tile) must equal the single-query result."""
    rng = np.random.default_rng(nq * 100 + dim)
    n = 1_100_000
    data = rng.random((n, dim), dtype=np.float32)
    queries = rng.random((nq, dim), dtype=np.float32)
    for dist, largest in ((D.Dot, True), (D.L2, False)):
        enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, dist, False))
        ids, sc = enc.topk_batch(enc.encode_query_batch(queries), 40, largest=largest)
        for qi in (0, nq - 1):
            wi, ws = enc.topk(enc.encode_query(queries[qi]), 40, largest=largest)
            assert np.array_equal(ids[qi], wi), (dist, qi)
            assert np.array_equal(sc[qi].view(np.uint32), ws.view(np.uint32))


def test_batch_degenerate_store_constant_data():
    """All values equal: alpha = 0, multiplier = 0 (and 0/0 codes): the integer pre-filter has no
    usable bound, the batch runs on the first kernel; results still equal the single-query path."""
    n, dim, nq, k = 3000, 192, 130, 7
    data = np.full((n, dim), 0.25, dtype=np.float32)
    queries = np.random.default_rng(2).random((nq, dim), dtype=np.float32)
    enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, D.Dot, False))
    b = enc.encode_query_batch(queries)
    got = enc.score_batch(b)
    ids, sc = enc.topk_batch(b, k)
    for qi in (0, 64, nq - 1):
        qobj = enc.encode_query(queries[qi])
        assert_bits_equal(got[qi], enc.score_all(qobj), f"query {qi}")
        want_ids, want_sc = enc.topk(qobj, k)
        assert np.array_equal(ids[qi], want_ids) and np.array_equal(sc[qi].view(np.uint32), want_sc.view(np.uint32))


def test_ping_pong_kernel_repeated_runs_are_bit_stable():
    """The ping-pong kernel orders its LDS-DMA ring with counted waits and raw barriers; an
    ordering bug would show as rare, timing-dependent wrong scores.  One store, many launches with
    fresh queries, every score of several queries compared on the device with the single-query scan
    (both tile shapes: 300 queries -> 256 x 256 tiles, 70 queries -> 128 x 512 tiles)."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    n, dim = 700_000, 448  # 7 K-tiles per row, 2735 row tiles: every workgroup walks several tiles
    data = torch.rand((n, dim), generator=g, device=dev)
    enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, D.L2, False))
    del data
    single = torch.empty(n, device=dev)
    for nq in (300, 70):
        out = torch.empty((nq, n), device=dev)
        for it in range(12):
            queries = torch.rand((nq, dim), generator=g, device=dev)
            enc.score_batch(enc.encode_query_batch(queries), out=out.view(-1))
            for qi in (0, nq // 2, nq - 1):
                enc.score_all(enc.encode_query(queries[qi]), out=single)
                assert torch.equal(out[qi].view(torch.int32), single.view(torch.int32)), (nq, it, qi)


def test_ping_pong_filter_repeated_runs_match_exact_topk():
    """Same idea for the filter pass (pre-filter in the accumulators, wave-private lists, scatter,
    per-query sort): repeated launches with fresh queries against the exact single-query top-k."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    n, dim, k = 1_100_000, 320, 20
    data = torch.rand((n, dim), generator=g, device=dev) - 0.3
    enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, D.Dot, False))
    del data
    for nq in (300, 70):
        for it in range(6):
            queries = torch.rand((nq, dim), generator=g, device=dev) - 0.3
            ids, sc = enc.topk_batch(enc.encode_query_batch(queries), k, largest=(it % 2 == 0))
            for qi in (0, nq // 3, nq - 1):
                want_ids, want_sc = enc.topk(enc.encode_query(queries[qi]), k, largest=(it % 2 == 0))
                assert np.array_equal(ids[qi], want_ids), (nq, it, qi)
                assert np.array_equal(sc[qi].view(np.uint32), want_sc.view(np.uint32)), (nq, it, qi)


@pytest.mark.parametrize("n,dim,nq", [(40_000, 768, 33), (100_003, 192, 130), (700_001, 128, 300), (33_000, 1536, 5)])
@pytest.mark.parametrize("largest", [True, False])
def test_topk_batch_medium_stores_take_the_matrix_core_path(n, dim, nq, largest):
    """Stores of 32k .. 1M rows (typical Qdrant segments) with many queries: one matrix-core pass for
    the whole batch instead of a launch chain per query; every list equals the exact single-query one."""
    rng = np.random.default_rng(n + nq)
    data = rng.random((n, dim), dtype=np.float32)
    dist = D.Dot if largest else D.L2
    enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, dist, False))
    queries = rng.random((nq, dim), dtype=np.float32)
    ids, sc = enc.topk_batch(enc.encode_query_batch(queries), 30, largest=largest)
    qobj = None
    for qi in range(nq):
        qobj = enc.encode_query(queries[qi], reuse=qobj)
        wi, ws = enc.topk(qobj, 30, largest=largest)
        assert np.array_equal(ids[qi], wi), (qi, n)
        assert np.array_equal(sc[qi].view(np.uint32), ws.view(np.uint32)), (qi, n)


def test_topk_batch_l1_large_store_runs_back_to_back_fused_scans():
    """L1 has no matrix form; above 2M rows the batch call enqueues the per-query fused scans back to
    back (one status read-back per 32 queries) — results equal the single-query call."""
    torch = pytest.importorskip("torch")
    n, dim, nq = 2_300_000, 64, 35
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    data = torch.rand((n, dim), generator=g, device="cuda")
    enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, D.L1, False))
    queries = np.random.default_rng(3).random((nq, dim), dtype=np.float32)
    ids, sc = enc.topk_batch(enc.encode_query_batch(queries), 20, largest=False)
    for qi in (0, 1, 17, 31, 32, 34):
        wi, ws = enc.topk(enc.encode_query(queries[qi]), 20, largest=False)
        assert np.array_equal(ids[qi], wi) and np.array_equal(sc[qi].view(np.uint32), ws.view(np.uint32)), qi


@pytest.mark.parametrize("dim,nq,dist", [(192, 4, D.Dot), (768, 3, D.Dot), (768, 2, D.L2), (1024, 4, D.Dot), (1024, 3, D.L1),
                                         (2000, 2, D.Dot), (768, 7, D.L1), (768, 8, D.Dot)])
def test_handful_of_queries_take_the_vector_alu_multi_query_scan(dim, nq, dist, qo):
    """2 .. 4 queries (any number for L1) over a store above 2M rows: passes of u8_scan_multi_kernel
    (queries in registers, rows streamed once per pass; 3 queries ride in a 4-wide pass) for score_batch
    and for the filtering scan of topk_batch — bit-identical to the single-query calls and to the oracle
    on sampled rows.  (8 Dot queries take the matrix-core path: same check.)"""
    torch = pytest.importorskip("torch")
    n = 2_200_003
    g = torch.Generator(device="cuda")
    g.manual_seed(dim + nq)
    data = torch.rand((n, dim), generator=g, device="cuda")
    enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, dist, False))
    md = enc.metadata
    queries = np.random.default_rng(dim).random((nq, dim), dtype=np.float32)
    batch = enc.encode_query_batch(queries)
    out = torch.empty(nq * n, dtype=torch.float32, device="cuda")
    enc.score_batch(batch, out=out)
    largest = dist == D.Dot
    ids, sc = enc.topk_batch(batch, 30, largest=largest)
    torch.cuda.synchronize()
    sb = out.view(nq, n)
    rows_idx = torch.randint(0, n, (2000,), generator=g, device="cuda")
    o_rows, o_meta = qo.u8_encode_with(data[rows_idx].cpu().numpy(), int(dist), False, float(md["alpha"]), float(md["offset"]))
    for qi in range(nq):
        q = enc.encode_query(queries[qi])
        single = enc.score_all(q, out=torch.empty(n, dtype=torch.float32, device="cuda"))
        assert torch.equal(sb[qi], single), f"score_batch query {qi}"
        wi, ws = enc.topk(q, 30, largest=largest)
        assert np.array_equal(ids[qi], wi) and np.array_equal(sc[qi].view(np.uint32), ws.view(np.uint32)), qi
        codes, qoff = qo.u8_encode_query(o_meta, queries[qi])
        order = qo.ORDER_AVX2 if md["actual_dim"] <= 1040 else qo.ORDER_SIMPLE
        assert_bits_equal(sb[qi][rows_idx].cpu().numpy(), qo.u8_score_all(o_meta, o_rows, codes, qoff, order=order),
                          f"query {qi} vs oracle")


# ---- row-streaming kernel (u8_gemm_rs_kernel): batches whose query tile fits in LDS
@pytest.mark.parametrize("n,dim,nq", [
    (70_001, 96, 20),      # one K-block per row (odd count), 32-query tile
    (50_000, 200, 64),     # row length 208: the last K-block runs into the next row (query image is zero there)
    (300_017, 384, 100),   # three K-blocks, 128-query tile, more than one row tile per workgroup, ragged tail
    (140_000, 512, 300),   # three 128-query tiles per row lane (the last one partly filled)
    (20_000, 1000, 128),   # eight K-blocks, row length not a multiple of 128
    (9_000, 1152, 128),    # the longest row a 128-query tile holds
    (9_000, 1168, 128),    # one step longer: only 64-query tiles fit, two would be needed -> ping-pong kernel
    (6_000, 2304, 33),     # the longest row a 64-query tile holds
    (3_000, 4600, 32),     # 32-query tile
    (3_000, 4600, 40),     # ... two of them would be needed -> ping-pong kernel
    (2_000, 4700, 8),      # no tile fits: the ping-pong kernel serves it
])
def test_row_streaming_kernel_shapes(n, dim, nq, qo):
    rng = np.random.default_rng(n + dim + nq)
    data = rng.random((n, dim), dtype=np.float32)
    queries = rng.random((nq, dim), dtype=np.float32)
    for dist, invert, largest in ((D.Dot, False, True), (D.L2, True, False)):
        enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, dist, invert))
        batch = enc.encode_query_batch(queries)
        got = enc.score_batch(batch)
        rows, meta = qo.u8_encode(data, int(dist), invert)
        for qi in sorted({0, nq // 2, nq - 1}):
            codes, qoff = qo.u8_encode_query(meta, queries[qi])
            want = qo.u8_score_all(meta, rows, codes, qoff, order=qo.ORDER_SIMPLE)
            assert_bits_equal(got[qi], want, f"{dist} query {qi} vs oracle")
        if n < 32768:
            continue  # top-k of small stores takes the exact single-query path
        ids, sc = enc.topk_batch(batch, 30, largest=largest)
        qobj = None
        for qi in sorted({0, 1, nq // 3, nq // 2, nq - 2, nq - 1}):
            qobj = enc.encode_query(queries[qi], reuse=qobj)
            wi, ws = enc.topk(qobj, 30, largest=largest)
            assert np.array_equal(ids[qi], wi), (qi, n, dim)
            assert np.array_equal(sc[qi].view(np.uint32), ws.view(np.uint32)), (qi, n, dim)


def test_row_streaming_kernel_pivot_sample_is_cached_and_stable():
    """The pivot sample is gathered once per handle; later calls (other batch sizes, k, direction)
    use prefixes of it and must keep returning the exact lists."""
    rng = np.random.default_rng(77)
    n, dim = 150_000, 256
    data = rng.random((n, dim), dtype=np.float32)
    enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, D.Dot, False))
    for nq, k, largest in ((7, 10, True), (90, 200, False), (300, 30, True), (7, 10, True)):
        queries = rng.random((nq, dim), dtype=np.float32)
        ids, sc = enc.topk_batch(enc.encode_query_batch(queries), k, largest=largest)
        for qi in (0, nq - 1):
            wi, ws = enc.topk(enc.encode_query(queries[qi]), k, largest=largest)
            assert np.array_equal(ids[qi], wi) and np.array_equal(sc[qi].view(np.uint32), ws.view(np.uint32)), (nq, qi)


def test_row_streaming_kernel_forced_for_many_query_tiles():
    """QAMD_GEMM_CFG=r (developer switch, read once per process) sends every batch whose tile fits
    through the row-streaming kernel: several query tiles per row lane, rows shared through the XCD's L2."""
    import os
    import subprocess
    import sys
    code = r'''
import numpy as np, quantization_amd as qa
D = qa.DistanceType
rng = np.random.default_rng(5)
n, dim, nq = 200_003, 192, 300
data = rng.random((n, dim), dtype=np.float32)
queries = rng.random((nq, dim), dtype=np.float32)
enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, D.L2, False))
b = enc.encode_query_batch(queries)
ids, sc = enc.topk_batch(b, 30, largest=False)
got = enc.score_batch(b)
for qi in (0, 127, 128, 255, 256, 299):
    q = enc.encode_query(queries[qi])
    wi, ws = enc.topk(q, 30, largest=False)
    assert np.array_equal(ids[qi], wi) and np.array_equal(sc[qi].view(np.uint32), ws.view(np.uint32)), qi
    assert np.array_equal(got[qi].view(np.uint32), enc.score_all(q).view(np.uint32)), qi
print("ok")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, QAMD_GEMM_CFG="r", PYTHONPATH=root)
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert res.returncode == 0 and "ok" in res.stdout, (res.stdout + res.stderr)[-2000:]


# ---- query-streaming kernels (u8_gemm_qs16_kernel on 16x16x64 MFMAs for rows of up to 1024 code bytes, u8_gemm_qs_kernel
# on 32x32x32 up to 1536): from 385 / 257 / 129 queries (rows of up to 384 / 768 / 1536 bytes: qs_min_queries); batches of up
# to 256 queries in chunks of 32 per wave
@pytest.mark.parametrize("n,dim,nq", [
    (140_000, 768, 300),    # stores of 131072+ rows take the query-streaming kernels by default: 5 chunks of 64 queries
    (140_000, 1536, 200),   # ... 96 resident rows, chunks of 32 queries
    (33_000, 768, 130),     # queries in registers (u8_gemm_qr16_kernel): 129 .. 256 queries on rows of 256 / 512 / 768 / 1024 bytes
    (70_001, 768, 256),     # ... all eight waves busy, several row blocks per workgroup, ragged tail
    (33_000, 512, 200),
    (33_000, 256, 129),
    (33_000, 384, 250),     # 384-byte rows on a 512-byte LDS pitch (places past the row's last chunk are filler)
    (33_000, 1024, 70),     # (from 65 queries on 1024-byte rows)
    (33_000, 1024, 129),
    (33_000, 1024, 256),    # 8 chunks of 32: the last batch size of the small-chunk form
    (33_000, 1536, 200),    # 96 resident rows, chunks of 32
    (33_000, 768, 257),     # the first batch size past two row-streaming tiles: 5 query chunks, three waves idle
    (33_000, 768, 256),     # ... and the last one the row-streaming kernel keeps
    (33_000, 768, 385),
    (33_000, 512, 390),
    (33_000, 1024, 700),    # the longest row of the 16x16x64 form: 16 k-steps (one left over after five turns of three)
    (33_000, 896, 300),     # 14 k-steps (two left over)
    (33_000, 640, 300),     # rows that end inside a 256-byte LDS group (pitch 768)
    (33_000, 256, 400),     # 4 k-steps
    (33_000, 1040, 300),    # one step longer: the 32x32x32 form
    (40_003, 96, 704),      # one K-block per row (odd count)
    (35_000, 200, 800),     # row length 208, two K-blocks; 13 query chunks over 8 waves
    (70_001, 384, 1024),    # three K-blocks (odd), two chunks per wave, several row blocks per workgroup, ragged tail
    (33_000, 1152, 720),    # the longest row a 128-row resident block holds
    (33_000, 1168, 720),    # one step longer: 96 resident rows
    (40_000, 1536, 800),    # the longest row 96 resident rows hold
    (33_000, 1552, 704),    # one step longer: ping-pong kernel
    (34_000, 128, 2100),    # two launch slices of 2048 queries
])
def test_query_streaming_kernel_shapes(n, dim, nq, qo):
    rng = np.random.default_rng(n + dim + nq)
    data = rng.random((n, dim), dtype=np.float32)
    queries = rng.random((nq, dim), dtype=np.float32)
    for dist, invert, largest in ((D.Dot, False, True), (D.L2, True, False)):
        enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, dist, invert))
        batch = enc.encode_query_batch(queries)
        picks = sorted({0, 1, 63, 64, nq // 3, nq // 2, nq - 65, nq - 2, nq - 1})
        if dist == D.Dot:  # scores of the whole batch: a few queries against the oracle
            got = enc.score_batch(batch)
            rows, meta = qo.u8_encode(data, int(dist), invert)
            for qi in picks[::3]:
                codes, qoff = qo.u8_encode_query(meta, queries[qi])
                want = qo.u8_score_all(meta, rows, codes, qoff, order=qo.ORDER_SIMPLE)
                assert_bits_equal(got[qi], want, f"{dist} query {qi} vs oracle")
            del got
        ids, sc = enc.topk_batch(batch, 30, largest=largest)
        qobj = None
        for qi in picks:
            qobj = enc.encode_query(queries[qi], reuse=qobj)
            wi, ws = enc.topk(qobj, 30, largest=largest)
            assert np.array_equal(ids[qi], wi), (qi, n, dim)
            assert np.array_equal(sc[qi].view(np.uint32), ws.view(np.uint32)), (qi, n, dim)


# ---- queries resident, rows streamed (late round 4): the filter pass of 129+ queries on stores of 131072+ rows of 256 / 384 / 512 /
# 768 bytes; the batch in groups of query tiles (as many as fit a CU's LDS: 12 / 18 / 24 / 38 tiles) that run side by side on the CUs
# of an XCD.  768-byte rows: u8_gemm_rk16_kernel (K loop outside, all tiles' accumulators in registers: 8, 10 or 12 tiles per
# group, padded), one to four groups where measured faster; shorter rows: u8_gemm_rq16_kernel (tiles outside), up to four groups
@pytest.mark.parametrize("n,dim,nq", [
    (140_001, 768, 129),    # one group of 10 tiles (the second half of the last tile pair is padding)
    (140_001, 768, 192),    # one full group: 12 tiles, 144 KiB of fragments
    (140_001, 768, 193),    # (two groups would be needed: the queries-in-registers kernel keeps 193 .. 256)
    (140_001, 768, 257),    # two groups of 10 and 8 tiles (both as 10: the smaller one padded with a tile pair that never passes)
    (150_000, 768, 384),    # two full groups
    (131_072, 768, 385),    # (three groups up to 512 queries: the query-streaming kernel keeps those)
    (140_001, 768, 545),    # three groups of 12 / 12 / 12 tiles (36 for 35), ten row streams per XCD (two CUs of 32 idle)
    (140_001, 768, 640),    # four groups of 10 tiles
    (140_001, 768, 768),    # four full groups: the largest batch of 768-byte rows it takes
    (140_001, 768, 769),
    (140_001, 512, 700),    # eight k-steps: three groups of 16 / 14 / 14 tiles
    (140_001, 512, 1152),   # four full groups of 18 tiles
    (140_001, 384, 300),    # six k-steps, one group
    (200_003, 256, 400),    # four k-steps, one group of 26 tiles
    (140_001, 256, 2400),   # four groups of 38 / 38 / 38 / 36 tiles
    (140_001, 760, 300),    # 760 code bytes + 8 of padding: the pad bytes are zero in rows and queries
])
def test_resident_queries_kernel_shapes(n, dim, nq):
    rng = np.random.default_rng(n + dim + nq)
    data = rng.random((n, dim), dtype=np.float32)
    queries = rng.random((nq, dim), dtype=np.float32)
    queries[nq // 2] = queries[0]  # the same query twice, in different groups at most sizes
    for dist, invert, largest in ((D.Dot, False, True), (D.L2, True, False), (D.Dot, True, False)):
        enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, dist, invert))
        ids, sc = enc.topk_batch(enc.encode_query_batch(queries), 30, largest=largest)
        assert np.array_equal(ids[nq // 2], ids[0]) and np.array_equal(sc[nq // 2], sc[0])
        qobj = None
        for qi in sorted(set(range(0, nq, 37)) | {1, 15, 16, nq - 2, nq - 1}):
            qobj = enc.encode_query(queries[qi], reuse=qobj)
            wi, ws = enc.topk(qobj, 30, largest=largest)
            assert np.array_equal(ids[qi], wi), (qi, n, dim)
            assert np.array_equal(sc[qi].view(np.uint32), ws.view(np.uint32)), (qi, n, dim)


def test_resident_queries_kernel_forced_for_many_groups():
    """QAMD_GEMM_CFG=s (a developer switch: only the tools/lib build reads it) sends batches of up to eight LDS images through
    the resident-queries kernels - 1024 queries of 768 bytes are six groups, 1536 eight, 200 two of 8 tiles: the same ids and
    score bits as the product library's own selection (the query-streaming / queries-in-registers kernels)."""
    import hashlib
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dev_lib = os.path.join(root, "tools", "lib", "libquantization_amd_dev.so")
    if not os.path.exists(dev_lib):
        pytest.skip("developer build not present")
    code = r"""
import sys, hashlib
sys.path.insert(0, %r)
import numpy as np
import quantization_amd as qa
D = qa.DistanceType
h = hashlib.sha256()
for n, dim, nq in ((140_001, 768, 1024), (131_072, 768, 1536), (140_001, 512, 2000), (140_001, 768, 200)):
    rng = np.random.default_rng(n + nq)
    data = rng.random((n, dim), dtype=np.float32)
    queries = rng.random((nq, dim), dtype=np.float32)
    enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, D.Dot, False))
    for largest in (True, False):
        ids, sc = enc.topk_batch(enc.encode_query_batch(queries), 30, largest=largest)
        h.update(np.asarray(ids).tobytes()); h.update(np.asarray(sc).tobytes())
print("DIGEST", h.hexdigest())
""" % root
    outs = []
    for env_extra in ({}, {"QAMD_LIB_PATH": dev_lib, "QAMD_GEMM_CFG": "s"}):
        env = dict(os.environ, **env_extra)
        res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
        assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
        outs.append([ln for ln in res.stdout.splitlines() if ln.startswith("DIGEST")][0])
    assert outs[0] == outs[1], outs


def test_query_streaming_kernel_reused_batch_object():
    """A batch object re-encoded with other queries (same shape) rebuilds its fragment-order copy."""
    rng = np.random.default_rng(11)
    n, dim, nq = 50_000, 256, 768
    data = rng.random((n, dim), dtype=np.float32)
    enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, D.Dot, False))
    batch = None
    for rep in range(2):
        queries = rng.random((nq, dim), dtype=np.float32)
        batch = enc.encode_query_batch(queries, reuse=batch)
        ids, sc = enc.topk_batch(batch, 10)
        for qi in (0, 400, nq - 1):
            wi, ws = enc.topk(enc.encode_query(queries[qi]), 10)
            assert np.array_equal(ids[qi], wi) and np.array_equal(sc[qi].view(np.uint32), ws.view(np.uint32)), (rep, qi)


def test_topk_batch_more_than_4096_queries():
    """Above 4096 queries the per-candidate scatter kernel serves (the grouped one keeps two words per query
    in LDS), the query-streaming kernel runs three slices of 2048 queries and k = 100 takes the sorting emit."""
    rng = np.random.default_rng(17)
    n, dim, nq = 60_000, 64, 4200
    data = rng.random((n, dim), dtype=np.float32)
    queries = rng.random((nq, dim), dtype=np.float32)
    enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, D.Dot, False))
    batch = enc.encode_query_batch(queries)
    for k in (30, 100):
        ids, sc = enc.topk_batch(batch, k)
        qobj = None
        for qi in (0, 2047, 2048, 4095, 4096, nq - 1):
            qobj = enc.encode_query(queries[qi], reuse=qobj)
            wi, ws = enc.topk(qobj, k)
            assert np.array_equal(ids[qi], wi), (k, qi)
            assert np.array_equal(sc[qi].view(np.uint32), ws.view(np.uint32)), (k, qi)


@pytest.mark.parametrize("nq", [100, 1024])
def test_topk_batch_of_identical_queries(nq):
    """Duplicate queries in one batch: every passing row appends to all their lists at once (bursts in the
    wave-private candidate lists); results must be exact and identical for every copy."""
    rng = np.random.default_rng(23)
    n, dim = 200_000, 128
    data = rng.random((n, dim), dtype=np.float32)
    enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, D.Dot, False))
    q = rng.random(dim, dtype=np.float32)
    ids, sc = enc.topk_batch(enc.encode_query_batch(np.repeat(q[None, :], nq, axis=0)), 30)
    wi, ws = enc.topk(enc.encode_query(q), 30)
    for qi in (0, nq // 2, nq - 1):
        assert np.array_equal(ids[qi], wi) and np.array_equal(sc[qi].view(np.uint32), ws.view(np.uint32)), qi
