"""Device top-k vs a numpy reference of the caller's heap (ann_benchmark_data.rs:151-167)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

qa = pytest.importorskip("quantization_amd")
D = qa.DistanceType


def _expect(scores, k, largest):
    idx = np.arange(scores.size)
    order = np.lexsort((idx, -scores if largest else scores))  # best first, ties -> lower id
    return order[:k].astype(np.uint32), scores[order[:k]]


@pytest.mark.parametrize("largest", [True, False])
@pytest.mark.parametrize("k", [1, 10, 30, 1024])
def test_u8_topk_matches_full_sort(k, largest):
    rng = np.random.default_rng(k)
    n, dim = 50000, 64
    data = rng.random((n, dim), dtype=np.float32)
    enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, D.Dot, False))
    q = enc.encode_query(rng.random(dim, dtype=np.float32))
    scores = enc.score_all(q)
    ids, sc = enc.topk(q, k, largest=largest)
    wi, ws = _expect(scores, k, largest)
    assert np.array_equal(ids, wi)
    assert np.array_equal(sc, ws)


def test_binary_topk_heavy_ties_break_to_lower_id():
    rng = np.random.default_rng(1)
    n, dim = 30000, 64  # only 65 distinct scores: massive ties
    data = np.where(rng.random((n, dim)) < 0.5, -1.0, 1.0).astype(np.float32)
    enc = qa.EncodedVectorsBin.encode(data, qa.VectorParameters(dim, n, D.Dot, False))
    q = enc.encode_query(data[17])
    scores = enc.score_all(q)
    for k, largest in ((30, True), (200, False)):
        ids, sc = enc.topk(q, k, largest=largest)
        wi, ws = _expect(scores, k, largest)
        assert np.array_equal(ids, wi) and np.array_equal(sc, ws)
    assert enc.topk(q, 1)[0][0] == 17


def test_topk_fewer_rows_than_k_and_device_outputs():
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(2)
    data = rng.random((5, 32), dtype=np.float32)
    enc = qa.EncodedVectorsPQ.encode(data, qa.VectorParameters(32, 5, D.L2, False), 4)
    q = enc.encode_query(data[2])
    ids, sc = enc.topk(q, 8, largest=False)
    assert ids[0] == 2 and sc[0] == 0.0
    assert np.all(ids[5:] == 0xFFFFFFFF) and np.all(np.isinf(sc[5:]))
    d_ids = torch.empty(8, dtype=torch.int32, device="cuda")
    d_sc = torch.empty(8, dtype=torch.float32, device="cuda")
    enc.topk(q, 8, largest=False, out_ids=d_ids, out_scores=d_sc)
    torch.cuda.synchronize()
    assert np.array_equal(d_ids.cpu().numpy().view(np.uint32), ids)


@pytest.mark.parametrize("largest", [True, False])
@pytest.mark.parametrize("k", [1, 30, 1024])
def test_fused_topk_large_store_u8(k, largest):
    """n >= 2^20 takes the fused path (sample pivot -> filtering scan -> one-workgroup sort)."""
    rng = np.random.default_rng(100 + k)
    n, dim = 1_200_000, 32
    data = rng.random((n, dim), dtype=np.float32)
    for dist in (D.Dot, D.L2):
        enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, dist, False))
        q = enc.encode_query(rng.random(dim, dtype=np.float32))
        scores = enc.score_all(q)
        ids, sc = enc.topk(q, k, largest=largest)
        wi, ws = _expect(scores, k, largest)
        assert np.array_equal(ids, wi)
        assert np.array_equal(sc.view(np.uint32), ws.view(np.uint32))


def test_fused_topk_binary_ties_and_fallback():
    rng = np.random.default_rng(5)
    for dim in (1024, 64):  # dim 64: 65 distinct scores, the pivot's tie group overflows -> exact fallback
        n = 1_500_000
        data = np.where(rng.random((n, dim)) < 0.5, -1.0, 1.0).astype(np.float32)
        enc = qa.EncodedVectorsBin.encode(data, qa.VectorParameters(dim, n, D.Dot, False))
        q = enc.encode_query(data[123])
        scores = enc.score_all(q)
        for k, largest in ((30, True), (1000, False)):
            ids, sc = enc.topk(q, k, largest=largest)
            wi, ws = _expect(scores, k, largest)
            assert np.array_equal(ids, wi) and np.array_equal(sc, ws), (dim, k)


def test_fused_topk_pq_and_sorted_adversarial_order():
    rng = np.random.default_rng(6)
    n, dim, chunk = 1_100_000, 128, 8
    cen = rng.random((256, dim), dtype=np.float32)
    rows = rng.integers(0, 256, size=(n, 16), dtype=np.uint8)
    enc = qa.EncodedVectorsPQ.from_storage(rows, qa.VectorParameters(dim, n, D.Dot, False), chunk, cen)
    q = enc.encode_query(rng.random(dim, dtype=np.float32))
    scores = enc.score_all(q)
    ids, sc = enc.topk(q, 100)
    wi, ws = _expect(scores, 100, True)
    assert np.array_equal(ids, wi) and np.array_equal(sc.view(np.uint32), ws.view(np.uint32))
    # rows stored in ascending-score order (a pessimal layout for a strided sample)
    data = np.sort(rng.random(n).astype(np.float32))[:, None] * np.ones((1, 16), dtype=np.float32)
    e2 = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(16, n, D.Dot, False))
    q2 = e2.encode_query(np.ones(16, dtype=np.float32))
    s2 = e2.score_all(q2)
    for largest in (True, False):
        ids, sc = e2.topk(q2, 50, largest=largest)
        wi, ws = _expect(s2, 50, largest)
        assert np.array_equal(ids, wi) and np.array_equal(sc, ws)


@pytest.mark.parametrize("n", [1, 7, 63, 64, 65, 1000, 33_333, 300_001])
def test_small_store_single_launch_topk_all_quantizers(n, qo):
    """The single-launch path (count <= 2M, k <= 64): u8 Dot/L1, binary (massive ties) and PQ, every k
    class, both directions, against a full stable sort of score_all."""
    rng = np.random.default_rng(n)
    dim = 96
    data = rng.random((n, dim), dtype=np.float32)
    stores = [
        qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, D.Dot, False)),
        qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, D.L1, True)),
        qa.EncodedVectorsBin.encode(data - 0.5, qa.VectorParameters(dim, n, D.Dot, False)),
        qa.EncodedVectorsPQ.encode(data, qa.VectorParameters(dim, n, D.L2, False), 8,
                                   centroids=rng.random((256, dim), dtype=np.float32)),
        qa.EncodedVectorsPQ.encode(data[:, :70].copy(), qa.VectorParameters(70, n, D.Dot, False), 4,  # m = 18: tail chunks
                                   centroids=rng.random((256, 70), dtype=np.float32)),
    ]
    for enc in stores:
        qdim = enc.vector_parameters.dim
        q = enc.encode_query(rng.random(qdim, dtype=np.float32) - (0.5 if isinstance(enc, qa.EncodedVectorsBin) else 0.0))
        scores = enc.score_all(q)
        for k, largest in ((1, True), (30, True), (30, False), (64, True), (5, False)):
            ids, sc = enc.topk(q, k, largest=largest)
            wi, ws = _expect(scores, k, largest)
            m = min(k, n)
            assert np.array_equal(ids[:m], wi[:m]), (type(enc).__name__, n, k, largest)
            assert np.array_equal(sc[:m].view(np.uint32), ws[:m].view(np.uint32))
            assert np.all(ids[m:] == 0xFFFFFFFF)


def test_small_store_topk_is_capturable_with_device_outputs():
    """No status read-back on the single-launch path: with device outputs the call only enqueues, so
    encode_query + topk replay from a hipGraph."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(4)
    n, dim = 50_000, 128
    data = rng.random((n, dim), dtype=np.float32)
    enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, D.Dot, False))
    qbuf = torch.from_numpy(rng.random(dim, dtype=np.float32)).cuda()
    d_ids = torch.empty(30, dtype=torch.int32, device="cuda")
    d_sc = torch.empty(30, dtype=torch.float32, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        qobj = enc.encode_query(qbuf)
        enc.topk(qobj, 30, out_ids=d_ids, out_scores=d_sc)  # warm-up: workspace allocation happens here
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            enc.encode_query(qbuf, reuse=qobj)
            enc.topk(qobj, 30, out_ids=d_ids, out_scores=d_sc)
    for trial in range(3):
        query = rng.random(dim, dtype=np.float32)
        qbuf.copy_(torch.from_numpy(query))
        g.replay()
        torch.cuda.synchronize()
        wi, ws = enc.topk(enc.encode_query(query), 30)
        assert np.array_equal(d_ids.cpu().numpy().view(np.uint32), wi), trial
        assert np.array_equal(d_sc.cpu().numpy().view(np.uint32), ws.view(np.uint32))


@pytest.mark.parametrize("dim", [65, 768, 896, 897, 1536])
@pytest.mark.parametrize("dist,invert", [(D.Dot, False), (D.L2, False), (D.L1, True), (D.Dot, True)])
def test_one_launch_search_quantises_the_host_query_in_the_topk_kernel(dim, dist, invert, qo):
    """encode_query(host query) is deferred (<= 896 code bytes); on a small store topk() is then ONE launch whose
    prologue quantises the query passed by value (u8_topk_small_fused_kernel).  Codes, offset and every score must
    be the reference's (encoded_vectors_u8.rs:290-329, :331-384), and the query object must afterwards behave as
    an ordinary encoded query.  897 / 1536 dims: not deferred, same results through the two-launch path."""
    from util import assert_bits_equal

    rng = np.random.default_rng(dim + int(dist) + 10 * invert)
    n = 20_000
    data = rng.random((n, dim), dtype=np.float32) - (0.5 if dist == D.L1 else 0.0)
    enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, dist, invert))
    rows, meta = qo.u8_encode(data, int(dist), invert)
    query = rng.random(dim, dtype=np.float32)
    query[::7] = [np.nan, np.inf, -np.inf, 1e30, -0.0][dim % 5]  # the encoder's edge values
    codes, qoff = qo.u8_encode_query(meta, query)
    want = qo.u8_score_all(meta, rows, codes, qoff, order=qo.ORDER_SIMPLE)
    for k, largest in ((30, True), (64, False), (1, True)):
        q = enc.encode_query(query)  # host query: nothing launched yet for <= 896 dims
        ids, sc = enc.topk(q, k, largest=largest)
        wi, ws = _expect(want, k, largest)
        assert np.array_equal(ids, wi), (k, largest)
        assert_bits_equal(sc, ws, "fused top-k scores")
        # workgroup 0 left the codes and the offset in the query object
        assert np.array_equal(q.encoded_query, codes)
        assert np.float32(q.offset).view(np.uint32) == np.float32(qoff).view(np.uint32)
        assert_bits_equal(enc.score_all(q), want, "score_all after the fused launch")
    # a deferred query whose FIRST consumer is not the small top-k is encoded on demand
    q = enc.encode_query(query)
    assert_bits_equal(enc.score_all(q), want, "score_all of a deferred query")
    q = enc.encode_query(query)
    assert np.array_equal(q.encoded_query, codes)
    q = enc.encode_query(query)
    assert np.float32(enc.score_point(q, 123)).view(np.uint32) == want[123:124].view(np.uint32)[0]
    q = enc.encode_query(query)
    ids_k, _ = enc.topk(q, 200, largest=True)  # k > 64: not the single-launch path
    assert np.array_equal(ids_k, _expect(want, 200, True)[0])
    # re-using the object for another query
    query2 = rng.random(dim, dtype=np.float32)
    q = enc.encode_query(query2, reuse=q)
    c2, o2 = qo.u8_encode_query(meta, query2)
    want2 = qo.u8_score_all(meta, rows, c2, o2, order=qo.ORDER_SIMPLE)
    ids, sc = enc.topk(q, 30)
    assert np.array_equal(ids, _expect(want2, 30, True)[0])
    assert_bits_equal(sc, _expect(want2, 30, True)[1], "reused query object")


def test_deferred_query_shared_by_threads():
    import threading

    rng = np.random.default_rng(77)
    n, dim = 50_000, 128
    enc = qa.EncodedVectorsU8.encode(rng.random((n, dim), dtype=np.float32), qa.VectorParameters(dim, n, D.Dot, False))
    query = rng.random(dim, dtype=np.float32)
    ref_q = enc.encode_query(query)
    want = enc.topk(ref_q, 20)
    want_all = enc.score_all(ref_q)
    for _ in range(20):
        q = enc.encode_query(query)  # deferred; four threads consume it at once
        out, errs = [None] * 4, []

        def work(i):
            try:
                out[i] = enc.topk(q, 20) if i % 2 == 0 else enc.score_all(q)
            except Exception as e:  # noqa: BLE001
                errs.append(e)

        ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert not errs, errs
        for i in range(4):
            if i % 2 == 0:
                assert np.array_equal(out[i][0], want[0]) and np.array_equal(out[i][1].view(np.uint32), want[1].view(np.uint32))
            else:
                assert np.array_equal(out[i].view(np.uint32), want_all.view(np.uint32))
