"""Batched (many queries at once) entry points of the PQ and binary quantizers — BASELINE config 4's
PQ leg and config 3's shape: every score and every top-k list must be bit-identical to looping the
single-query call (which is itself oracle-checked), and spot-checked against the oracle directly."""
import numpy as np
import pytest

from util import assert_bits_equal

pytestmark = pytest.mark.gpu

qa = pytest.importorskip("quantization_amd")
torch = pytest.importorskip("torch")
D = qa.DistanceType


@pytest.mark.parametrize("n,m_dim", [(120_000, (96, 8)), (3000, (70, 4))])
def test_pq_batch_equals_single_query_loop(n, m_dim, qo):
    dim, chunk = m_dim[0] * m_dim[1], m_dim[1]
    rng = np.random.default_rng(n)
    cen = rng.random((256, dim), dtype=np.float32)
    m = qa.EncodedVectorsPQ.get_quantized_vector_size(qa.VectorParameters(dim, n, D.Dot, False), chunk)
    rows = rng.integers(0, 256, (n, m), dtype=np.uint8)
    enc = qa.EncodedVectorsPQ.from_storage(rows, qa.VectorParameters(dim, n, D.Dot, False), chunk, cen)
    Q = 37
    queries = rng.random((Q, dim), dtype=np.float32)
    batch = enc.encode_query_batch(queries)
    sb = enc.score_batch(batch)
    ids, sc = enc.topk_batch(batch, 30)
    ids_s, sc_s = enc.topk_batch(batch, 17, largest=False)
    for qi in range(Q):
        q = enc.encode_query(queries[qi])
        assert_bits_equal(sb[qi], enc.score_all(q), f"score_batch query {qi}")
        wi, ws = enc.topk(q, 30)
        assert np.array_equal(ids[qi], wi) and np.array_equal(sc[qi].view(np.uint32), ws.view(np.uint32)), qi
        wi, ws = enc.topk(q, 17, largest=False)
        assert np.array_equal(ids_s[qi], wi) and np.array_equal(sc_s[qi].view(np.uint32), ws.view(np.uint32)), qi
    lut = qo.pq_encode_query(queries[5], chunk, cen, qo.DOT, False)
    assert_bits_equal(sb[5][:4000], qo.pq_score_all(rows[:4000], lut, order=qo.ORDER_SSE), "vs oracle")
    # device in / device out
    dq = torch.from_numpy(queries).cuda()
    b2 = enc.encode_query_batch(dq, reuse=batch)
    d_ids = torch.empty(Q * 30, dtype=torch.int32, device="cuda")
    d_sc = torch.empty(Q * 30, dtype=torch.float32, device="cuda")
    enc.topk_batch(b2, 30, out_ids=d_ids, out_scores=d_sc)
    torch.cuda.synchronize()
    assert np.array_equal(d_ids.cpu().numpy().view(np.uint32).reshape(Q, 30), ids)


@pytest.mark.parametrize("m,chunk,n,nq", [
    (96, 1, 1_100_003, 23),   # groups of 4 (five of them), 2 and one query alone; runs of four blocks per wave within a quarter of the machine
    (192, 1, 1_050_011, 9),   # rows of two LUT slices: four queries' lane sums side by side in the carry buffer, twice, then one alone
    (80, 1, 1_060_001, 8),    # padded ring rows
    (48, 2, 1_200_002, 6),    # two rows per ring row: a group of 4 and one of 2
    (100, 1, 1_048_579, 4),   # rows on a 112-byte pitch, zero table columns past m
])
def test_pq_batch_filter_passes_side_by_side(m, chunk, n, nq, qo):
    """topk_batch on a PQ store of a million rows and more: the filter passes of 4 / 2 queries run in ONE launch, every query's table
    in the LDS of its own CUs and the rows shared through L2 (SkewBatch) - ids and score bits of every query must be the single-query
    top-k's (whose scan is pinned to the oracle by tests/test_gpu_pq.py), both directions; one query also against the oracle's scores."""
    dim = m * chunk
    rng = np.random.default_rng(m * 7 + nq)
    cen = (rng.random((256, dim), dtype=np.float32) - 0.5).astype(np.float32)
    rows = rng.integers(0, 256, size=(n, m), dtype=np.uint8)
    queries = (rng.random((nq, dim), dtype=np.float32) - 0.5).astype(np.float32)
    for dist, invert, largest in ((D.Dot, False, True), (D.L2, True, False)):
        enc = qa.EncodedVectorsPQ.from_storage(rows, qa.VectorParameters(dim, n, dist, invert), chunk, cen)
        ids, sc = enc.topk_batch(enc.encode_query_batch(queries), 30, largest=largest)
        qobj = None
        for qi in range(nq):
            qobj = enc.encode_query(queries[qi], reuse=qobj)
            wi, ws = enc.topk(qobj, 30, largest=largest)
            assert np.array_equal(ids[qi], wi), (m, qi)
            assert np.array_equal(sc[qi].view(np.uint32), ws.view(np.uint32)), (m, qi)
        lut = qo.pq_encode_query(queries[nq - 1], chunk, cen, int(dist), invert)
        want = qo.pq_score_all(rows, lut, order=qo.ORDER_SSE)
        order = np.lexsort((np.arange(n), -want if largest else want))[:30]
        assert np.array_equal(sc[nq - 1].view(np.uint32), want[order].view(np.uint32)), "vs oracle"


@pytest.mark.parametrize("dim,n", [(1024, 200_000), (2048, 60_000), (4096, 40_000), (8192, 20_000), (256, 50_000), (65, 5000)])
def test_binary_batch_equals_single_query_loop(dim, n, qo):
    rng = np.random.default_rng(dim)
    data = np.where(rng.random((n, dim)) < 0.5, -1.0, 1.0).astype(np.float32)
    enc = qa.EncodedVectorsBin.encode(data, qa.VectorParameters(dim, n, D.Dot, False))
    Q = 23  # 8 + 8 + 4 + 2 + 1: every multi-query step size
    queries = np.where(rng.random((Q, dim)) < 0.5, -1.0, 1.0).astype(np.float32)
    batch = enc.encode_query_batch(queries)
    sb = enc.score_batch(batch)
    assert np.array_equal(sb, (queries @ data.T).astype(np.float32)), "binary Dot on +-1 data is the exact f32 dot"
    ids, sc = enc.topk_batch(batch, 30)
    for qi in (0, 7, 8, 15, 19, 21, 22):
        q = enc.encode_query(queries[qi])
        assert np.array_equal(sb[qi], enc.score_all(q))
        wi, ws = enc.topk(q, 30)
        assert np.array_equal(ids[qi], wi) and np.array_equal(sc[qi], ws), qi
    rows = qo.bin_encode(data[:3000])
    want = qo.bin_score_all(rows, qo.bin_encode(queries[3:4])[0], dim, qo.DOT, False)
    assert np.array_equal(sb[3][:3000], want)
    out = torch.empty(Q * n, dtype=torch.float32, device="cuda")
    enc.score_batch(batch, out=out)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().reshape(Q, n), sb)


def test_batch_edge_cases():
    rng = np.random.default_rng(1)
    dim = 64
    data = rng.standard_normal((10, dim)).astype(np.float32)
    enc = qa.EncodedVectorsBin.encode(data, qa.VectorParameters(dim, 10, D.L2, False))
    b = enc.encode_query_batch(data[:3])
    ids, sc = enc.topk_batch(b, 12, largest=False)
    assert ids.shape == (3, 12) and np.all(ids[:, 10:] == 0xFFFFFFFF) and ids[0, 0] == 0 and ids[2, 0] == 2
    cen = rng.random((256, dim), dtype=np.float32)
    penc = qa.EncodedVectorsPQ.encode(data, qa.VectorParameters(dim, 10, D.Dot, False), 8, centroids=cen)
    pb = penc.encode_query_batch(data[:2])
    assert penc.score_batch(pb).shape == (2, 10)
    with pytest.raises(qa.EncodingError):
        penc.encode_query_batch(rng.random((2, dim + 1), dtype=np.float32))


@pytest.mark.parametrize("dim,n,nq", [
    (1024, 100_003, 64),    # one 64-query tile, eight K-blocks
    (1000, 70_000, 100),    # dim not a multiple of 128: pad bits are zero in rows and queries; two tiles
    (128, 300_000, 40),     # one K-block, heavy ties (129 distinct scores): lists overflow -> exact path per query
    (2304, 40_000, 16),     # 18 K-blocks: three register passes per row
    (4992, 33_000, 33),     # 39 K-blocks: only a 32-query tile fits in LDS
    (640, 50_000, 300),     # five tiles
    (1024, 70_001, 130),    # 129+ queries on rows of 512 / 1024 bits: the FP4 kernel (bin_gemm_qs4_kernel), chunks of 32 queries
    (1024, 70_001, 600),    # ... chunks of 64 queries, several row blocks per workgroup, ragged tail
    (512, 50_000, 257),     # 512-bit rows: four k-steps
    (1000, 50_000, 200),    # pad bits inside a 1024-bit row
    (768, 60_000, 300),     # 768-bit rows: six k-steps, 384 bytes of nibbles on a 512-byte LDS pitch
    (700, 40_000, 140),     # ... with pad bits, chunks of 32 queries in registers
    (1536, 40_000, 300),    # 1536-bit rows: 96-row blocks, twelve k-steps
    (1500, 35_000, 100 + 60),  # ... pad bits, chunks of 32 queries in registers (12 k-steps x 2 tiles)
    (1024, 40_000, 2100),   # two launch slices of 2048 queries
    # round 4: batches whose nibble image fits in LDS take the row-streaming fp4 form (bin_gemm_rs4_kernel), from 3 / 5 queries on:
    (1024, 70_001, 12),     # one tile pair, the second tile all padding
    (1024, 50_001, 3), (1024, 50_001, 5), (768, 40_000, 7), (512, 40_000, 11), (1536, 40_000, 9),  # the smallest batches it takes
    (1024, 50_001, 4), (2048, 40_000, 5),  # (4 queries, other row lengths: the vector-ALU scan)
    (1024, 33_000, 288),    # the largest batch of 1024-bit rows (144 KiB of nibbles)
    (512, 40_000, 608),     # ... of 512-bit rows: 38 query tiles, four k-steps
    (1536, 50_000, 192),    # ... of 1536-bit rows: twelve k-steps (193 queries go to the query-streaming form)
    (768, 33_333, 16),      # six k-steps (8-byte row loads), a ragged last trip
    (1024, 32_768 + 31, 64),
    # ... and larger batches in up to four passes of that form, the queries spread evenly over the passes:
    (1024, 40_000, 289),    # two passes of 160
    (1024, 33_000, 1024),   # four of 256
    (1024, 33_000, 1152),   # four of 288: the largest; 1153 queries go to the query-streaming form
    (1024, 33_000, 1153),
    (768, 33_000, 900),     # three passes of 320 at 768 bits
])
def test_binary_batch_on_the_matrix_cores(dim, n, nq, qo):
    """3 and 5+ queries (12+ on other row lengths) on 32k rows and more take the matrix cores - bin_gemm_rs_kernel (bits expanded to 0/1 bytes in
    registers, int8 MFMA, u8-style epilogue with integer operands); on rows of 512 / 768 / 1024 / 1536 bits the FP4 matrix
    cores (bits as E2M1 nibbles, exact f32 counts): bin_gemm_rs4_kernel while the batch's nibble image fits in LDS (queries
    resident, every wave streams its own rows), bin_gemm_qs4_kernel beyond: every list must equal the single-query
    top-k AND the oracle's restatement of the caller loop (score_point for every row,
    encoded_vectors_binary.rs:293-300 -> calculate_metric :219-253, then a stable best-k: ties to the
    lower id), for the four metric variants and both directions."""
    rng = np.random.default_rng(dim + nq)
    data = rng.standard_normal((n, dim)).astype(np.float32)
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    rows = qo.bin_encode(data)
    qbits = qo.bin_encode(queries)
    for dist, invert, largest in ((D.Dot, False, True), (D.L2, False, False), (D.Dot, True, False), (D.L1, True, True)):
        enc = qa.EncodedVectorsBin.encode(data, qa.VectorParameters(dim, n, dist, invert))
        ids, sc = enc.topk_batch(enc.encode_query_batch(queries), 30, largest=largest)
        picks = sorted({0, 1, min(31, nq - 1), min(32, nq - 1), nq // 2, nq - 2, nq - 1} |
                       {q for b in (160, 256, 288, 320, 512, 576, 640, 768, 864) for q in (b - 1, b, b + 1) if q < nq})  # pass boundaries
        for j, qi in enumerate(picks):
            wi, ws = enc.topk(enc.encode_query(queries[qi]), 30, largest=largest)
            assert np.array_equal(ids[qi], wi), (dist, invert, largest, qi)
            assert np.array_equal(sc[qi].view(np.uint32), ws.view(np.uint32)), (dist, invert, largest, qi)
            if j % 3 == 0:  # the oracle: every score of the store, stable best-30
                want = qo.bin_score_all(rows, qbits[qi], dim, int(dist), invert)
                order = np.lexsort((np.arange(n), -want if largest else want))[:30]
                assert np.array_equal(ids[qi], order.astype(np.uint32)), ("vs oracle", dist, invert, largest, qi)
                assert np.array_equal(sc[qi].view(np.uint32), want[order].view(np.uint32)), ("vs oracle", dist, invert, qi)


def test_binary_batch_of_identical_queries():
    """Near-duplicate queries make every passing row append to all lists at once (bursts in one wave's
    candidate list): the result must still be exact, and the fast path must hold (no wave-list overflow)."""
    rng = np.random.default_rng(3)
    n, dim, nq = 400_000, 512, 64
    data = rng.standard_normal((n, dim)).astype(np.float32)
    enc = qa.EncodedVectorsBin.encode(data, qa.VectorParameters(dim, n, D.Dot, False))
    q = rng.standard_normal(dim).astype(np.float32)
    queries = np.repeat(q[None, :], nq, axis=0)
    ids, sc = enc.topk_batch(enc.encode_query_batch(queries), 30)
    wi, ws = enc.topk(enc.encode_query(q), 30)
    for qi in (0, 31, 63):
        assert np.array_equal(ids[qi], wi) and np.array_equal(sc[qi], ws), qi
