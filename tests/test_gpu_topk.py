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
