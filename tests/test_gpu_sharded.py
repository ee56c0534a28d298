"""Single-process row-sharded handles (csrc/sharded.hip, `qamd_*_sharded_*`): every result must be
bit-identical to the same call on ONE handle holding all rows — ids, score bits and the tie rule —
for 2, 3 and 8 shards.  One GPU: the shards are logical (devices = [0] * G), which covers the index
arithmetic, the worker fan-out and the device-side merge; the two-device test runs where a second
GPU exists."""
import numpy as np
import pytest

from util import assert_bits_equal

pytestmark = pytest.mark.gpu

qa = pytest.importorskip("quantization_amd")
torch = pytest.importorskip("torch")
D = qa.DistanceType


def _same_topk(a, b):
    assert np.array_equal(a[0], b[0]), "ids"
    assert_bits_equal(a[1], b[1], "top-k scores")


@pytest.mark.parametrize("G", [2, 3, 8])
@pytest.mark.parametrize("dist", [D.Dot, D.L2])
def test_u8_sharded_equals_single_handle(G, dist, qo):
    rng = np.random.default_rng(100 + G)
    n, dim = 300_007, 72  # ragged shard sizes; >= 32768 rows per shard at G = 8 -> fused top-k inside the shards
    data = rng.random((n, dim), dtype=np.float32)
    vp = qa.VectorParameters(dim, n, dist, False)
    one = qa.EncodedVectorsU8.encode(data, vp)
    sh = qa.ShardedVectorsU8.encode(data, vp, [0] * G)
    assert sh.n_shards == G
    m1, ms = one.metadata, sh.metadata
    for key in ("alpha", "offset", "multiplier"):
        assert np.float32(m1[key]).view(np.uint32) == np.float32(ms[key]).view(np.uint32), key
    rows = one.storage_bytes()
    for g in range(G):
        shard, base = sh.shard(g)
        b, e = sh.shard_range(g)
        assert base == b and shard.count == e - b
        assert np.array_equal(shard.storage_bytes(), rows[b:e]), f"shard {g} rows"
    query = rng.random(dim, dtype=np.float32)
    q1, qs = one.encode_query(query), sh.encode_query(query)
    s1 = one.score_all(q1)
    assert_bits_equal(sh.score_all(qs), s1, "sharded score_all (host out)")
    d_out = torch.empty(n, dtype=torch.float32, device="cuda:0")
    sh.score_all(qs, out=d_out)
    assert_bits_equal(d_out.cpu().numpy(), s1, "sharded score_all (device out)")
    for k, largest in ((30, True), (30, False), (1000, True), (1, False)):
        _same_topk(sh.topk(qs, k, largest=largest), one.topk(q1, k, largest=largest))
    # oracle, so that "equal to the single handle" is anchored to the reference arithmetic too
    o_rows, o_meta = qo.u8_encode(data[:4096], int(dist), False)  # not the same alpha: own check below
    o_rows, o_meta = qo.u8_encode_with(data[:4096], int(dist), False, float(m1["alpha"]), float(m1["offset"]))
    codes, qoff = qo.u8_encode_query(o_meta, query)
    assert_bits_equal(s1[:4096], qo.u8_score_all(o_meta, o_rows, codes, qoff, order=qo.ORDER_AVX2), "vs oracle")
    # device query, device outputs
    dq = torch.from_numpy(query).cuda()
    qd = sh.encode_query(dq)
    ids_d = torch.empty(30, dtype=torch.int32, device="cuda:0")
    sc_d = torch.empty(30, dtype=torch.float32, device="cuda:0")
    sh.topk(qd, 30, out_ids=ids_d, out_scores=sc_d)
    torch.cuda.synchronize()
    _same_topk((ids_d.cpu().numpy().view(np.uint32), sc_d.cpu().numpy()), one.topk(q1, 30))


def test_u8_sharded_quantile_from_rows_and_heavy_ties(qo):
    rng = np.random.default_rng(7)
    n, dim = 60_000, 32
    data = rng.random((n, dim), dtype=np.float32)
    vp = qa.VectorParameters(dim, n, D.Dot, True)
    one = qa.EncodedVectorsU8.encode(data, vp, quantile=0.99)
    sh = qa.ShardedVectorsU8.encode(data, vp, [0, 0, 0], quantile=0.99)
    rows = one.storage_bytes()
    o_rows, o_meta = qo.u8_encode(data, qo.DOT, True, quantile=0.99)
    assert np.array_equal(rows, o_rows)
    got = np.concatenate([sh.shard(g)[0].storage_bytes() for g in range(3)])
    assert np.array_equal(got, rows)
    # adopt reference-format rows (what a store encoded by the crate looks like), 5 shards
    sh2 = qa.ShardedVectorsU8.from_storage(rows, one.metadata, [0] * 5)
    # few distinct values -> massive score ties: the merge must keep the lower global id
    tied = np.repeat(rng.random((50, dim), dtype=np.float32), 1200, axis=0)  # 60000 rows, 50 distinct
    vt = qa.VectorParameters(dim, n, D.L2, False)
    one_t = qa.EncodedVectorsU8.encode(tied, vt)
    sh_t = qa.ShardedVectorsU8.encode(tied, vt, [0] * 7)
    query = rng.random(dim, dtype=np.float32)
    for (a, b) in ((sh2, one), (sh_t, one_t)):
        qa_, qb = a.encode_query(query), b.encode_query(query)
        assert_bits_equal(a.score_all(qa_), b.score_all(qb), "score_all")
        for k, largest in ((30, True), (500, False)):
            _same_topk(a.topk(qa_, k, largest=largest), b.topk(qb, k, largest=largest))


@pytest.mark.parametrize("G", [2, 8])
def test_binary_sharded_equals_single_handle(G):
    rng = np.random.default_rng(G)
    n, dim = 280_001, 256
    data = np.where(rng.random((n, dim)) < 0.5, -1.0, 1.0).astype(np.float32)
    vp = qa.VectorParameters(dim, n, D.Dot, False)
    one = qa.EncodedVectorsBin.encode(data, vp)
    sh = qa.ShardedVectorsBin.encode(data, vp, [0] * G)
    rows = one.storage_bytes()
    for g in range(G):
        b, e = sh.shard_range(g)
        assert np.array_equal(sh.shard(g)[0].storage_bytes(), rows[b:e])
    query = data[12345]
    q1, qs = one.encode_query(query), sh.encode_query(query)
    s1 = one.score_all(q1)
    assert np.array_equal(sh.score_all(qs), s1)
    assert np.array_equal(s1, (data @ query).astype(np.float32))  # the reference's known-answer property
    for k, largest in ((30, True), (64, False)):  # only 257 distinct scores: ties everywhere
        _same_topk(sh.topk(qs, k, largest=largest), one.topk(q1, k, largest=largest))
    sh2 = qa.ShardedVectorsBin.from_storage(rows, vp, [0] * 3)
    _same_topk(sh2.topk(sh2.encode_query(query), 30), one.topk(q1, 30))


@pytest.mark.parametrize("G", [3, 8])
def test_pq_sharded_equals_single_handle(G, qo):
    rng = np.random.default_rng(40 + G)
    n, dim, chunk = 270_000, 64, 4
    data = rng.random((n, dim), dtype=np.float32)
    cen = rng.random((256, dim), dtype=np.float32)
    vp = qa.VectorParameters(dim, n, D.Dot, False)
    one = qa.EncodedVectorsPQ.encode(data, vp, chunk, centroids=cen)
    sh = qa.ShardedVectorsPQ.encode(data, vp, chunk, [0] * G, centroids=cen)
    rows = one.storage_bytes()
    assert np.array_equal(rows[:2000], qo.pq_encode(data[:2000], chunk, cen))
    for g in range(G):
        b, e = sh.shard_range(g)
        assert np.array_equal(sh.shard(g)[0].storage_bytes(), rows[b:e])
    query = rng.random(dim, dtype=np.float32)
    q1, qs = one.encode_query(query), sh.encode_query(query)
    s1 = one.score_all(q1)
    assert_bits_equal(sh.score_all(qs), s1, "pq sharded score_all")
    lut = qo.pq_encode_query(query, chunk, cen, qo.DOT, False)
    assert_bits_equal(s1[:2000], qo.pq_score_all(rows[:2000], lut, order=qo.ORDER_SSE), "vs oracle")
    for k, largest in ((30, True), (300, False)):
        _same_topk(sh.topk(qs, k, largest=largest), one.topk(q1, k, largest=largest))
    # trained centroids: one training for all shards == the single handle's training
    vp2 = qa.VectorParameters(dim, 20_000, D.L2, False)
    one2 = qa.EncodedVectorsPQ.encode(data[:20_000], vp2, chunk, max_kmeans_threads=2)
    sh2 = qa.ShardedVectorsPQ.encode(data[:20_000], vp2, chunk, [0] * G, max_kmeans_threads=2)
    assert np.array_equal(sh2.centroids.view(np.uint32), one2.centroids.view(np.uint32))
    got = np.concatenate([sh2.shard(g)[0].storage_bytes() for g in range(G)])
    assert np.array_equal(got, one2.storage_bytes())


def test_u8_sharded_topk_batch_equals_single_handle():
    rng = np.random.default_rng(9)
    n, dim, Q, k = 400_000, 192, 40, 30
    data = rng.random((n, dim), dtype=np.float32)
    vp = qa.VectorParameters(dim, n, D.Dot, False)
    one = qa.EncodedVectorsU8.encode(data, vp)
    queries = rng.random((Q, dim), dtype=np.float32)
    ids1, sc1 = one.topk_batch(one.encode_query_batch(queries), k)
    for G in (2, 5):
        sh = qa.ShardedVectorsU8.encode(data, vp, [0] * G)
        ids, sc = sh.topk_batch(sh.encode_query_batch(queries), k)
        assert np.array_equal(ids, ids1)
        assert_bits_equal(sc, sc1, f"topk_batch G={G}")
        # per query it is also the single-query sharded result
        qs = sh.encode_query(queries[3])
        _same_topk(sh.topk(qs, k), (ids1[3], sc1[3]))


@pytest.mark.parametrize("kind", ["bin", "pq"])
def test_bin_pq_sharded_topk_batch_equals_single_handle(kind):
    rng = np.random.default_rng(19)
    Q, k = 12, 30
    if kind == "bin":
        n, dim = 300_001, 512
        data = np.where(rng.random((n, dim)) < 0.5, -1.0, 1.0).astype(np.float32)
        vp = qa.VectorParameters(dim, n, D.Dot, False)
        one = qa.EncodedVectorsBin.encode(data, vp)
        make = lambda G: qa.ShardedVectorsBin.encode(data, vp, [0] * G)
        queries = data[rng.integers(0, n, Q)]
    else:
        n, dim, chunk = 250_000, 64, 2
        data = rng.random((n, dim), dtype=np.float32)
        cen = rng.random((256, dim), dtype=np.float32)
        vp = qa.VectorParameters(dim, n, D.L2, False)
        one = qa.EncodedVectorsPQ.encode(data, vp, chunk, centroids=cen)
        make = lambda G: qa.ShardedVectorsPQ.encode(data, vp, chunk, [0] * G, centroids=cen)
        queries = rng.random((Q, dim), dtype=np.float32)
    for largest in (True, False):
        ids1, sc1 = one.topk_batch(one.encode_query_batch(queries), k, largest=largest)
        for G in (2, 7):
            sh = make(G)
            batch = sh.encode_query_batch(queries)
            ids, sc = sh.topk_batch(batch, k, largest=largest)
            assert np.array_equal(ids, ids1), (kind, G, largest)
            assert_bits_equal(sc, sc1, f"{kind} topk_batch G={G}")
            # a batch object is reusable for a second set of queries of the same shape
            batch = sh.encode_query_batch(queries[::-1].copy(), reuse=batch)
            ids_r, sc_r = sh.topk_batch(batch, k, largest=largest)
            assert np.array_equal(ids_r, ids1[::-1])
            _same_topk(sh.topk(sh.encode_query(queries[5]), k, largest=largest), (ids1[5], sc1[5]))


def test_sharded_edge_cases_and_errors():
    rng = np.random.default_rng(10)
    dim = 16
    # more shards than rows: empty shards; k larger than the store
    data = rng.random((5, dim), dtype=np.float32)
    vp = qa.VectorParameters(dim, 5, D.Dot, False)
    one = qa.EncodedVectorsU8.encode(data, vp)
    sh = qa.ShardedVectorsU8.encode(data, vp, [0] * 8)
    q = rng.random(dim, dtype=np.float32)
    assert_bits_equal(sh.score_all(sh.encode_query(q)), one.score_all(one.encode_query(q)), "tiny store")
    ids, sc = sh.topk(sh.encode_query(q), 8)
    ids1, sc1 = one.topk(one.encode_query(q), 8)
    assert np.array_equal(ids, ids1) and np.array_equal(sc, sc1)
    assert np.all(ids[5:] == 0xFFFFFFFF)
    # count == 0
    empty = qa.ShardedVectorsU8.encode(np.zeros((0, dim), np.float32), qa.VectorParameters(dim, 0, D.Dot, False), [0, 0])
    assert empty.score_all(empty.encode_query(q)).size == 0
    # stop_condition
    big = rng.random((50_000, dim), dtype=np.float32)
    with pytest.raises(qa.EncodingError) as e:
        qa.ShardedVectorsU8.encode(big, qa.VectorParameters(dim, 50_000, D.Dot, False), [0, 0], stop_condition=lambda: True)
    assert e.value.stopped
    # bad device
    with pytest.raises(qa.EncodingError):
        qa.ShardedVectorsU8.encode(big, qa.VectorParameters(dim, 50_000, D.Dot, False), [0, 99])


@pytest.mark.skipif(qa.lib().qamd_device_count() < 2, reason="needs two GPUs")
def test_u8_sharded_on_two_real_devices():
    rng = np.random.default_rng(11)
    n, dim = 200_000, 128
    data = rng.random((n, dim), dtype=np.float32)
    vp = qa.VectorParameters(dim, n, D.Dot, False)
    one = qa.EncodedVectorsU8.encode(data, vp)
    sh = qa.ShardedVectorsU8.encode(data, vp, [0, 1, 1, 0])
    q = rng.random(dim, dtype=np.float32)
    q1, qs = one.encode_query(q), sh.encode_query(q)
    assert_bits_equal(sh.score_all(qs), one.score_all(q1), "two devices, host out")
    out = torch.empty(n, dtype=torch.float32, device="cuda:0")
    sh.score_all(qs, out=out)
    assert_bits_equal(out.cpu().numpy(), one.score_all(q1), "two devices, device out (peer copies)")
    _same_topk(sh.topk(qs, 30), one.topk(q1, 30))


def test_sharded_handle_concurrent_callers_get_their_own_answers():
    """Callers on several threads share one sharded handle with no per-handle lock (their jobs
    interleave on the shards' queues, each call leases its own exchange buffers): every caller must
    still get its own exact answer.  The timing side of this is tests/c_abi/sharded_threads.c."""
    import threading

    rng = np.random.default_rng(12)
    n, dim = 120_000, 64
    data = rng.random((n, dim), dtype=np.float32)
    vp = qa.VectorParameters(dim, n, D.Dot, False)
    one = qa.EncodedVectorsU8.encode(data, vp)
    sh = qa.ShardedVectorsU8.encode(data, vp, [0] * 4)
    queries = rng.random((6, dim), dtype=np.float32)
    want = [one.topk(one.encode_query(q), 20) for q in queries]
    errors = []

    def worker(i):
        try:
            for _ in range(15):
                qs = sh.encode_query(queries[i])
                ids, sc = sh.topk(qs, 20)
                if not (np.array_equal(ids, want[i][0]) and np.array_equal(sc.view(np.uint32), want[i][1].view(np.uint32))):
                    errors.append(i)
                s_all = sh.score_all(qs)
                if not np.array_equal(s_all.view(np.uint32), one.score_all(one.encode_query(queries[i])).view(np.uint32)):
                    errors.append(100 + i)
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_sharded_calls_are_ordered_after_the_callers_stream(qo):
    """The lanes run on private streams: a query (or a store) that a kernel on the caller's stream is
    still producing must not be read early.  The producer here is a long chain of
    element-wise kernels ending in the copy that makes the real values."""
    rng = np.random.default_rng(21)
    n, dim = 50_000, 128
    data = rng.random((n, dim), dtype=np.float32)
    vp = qa.VectorParameters(dim, n, D.Dot, False)
    one = qa.EncodedVectorsU8.encode(data, vp)
    sh = qa.ShardedVectorsU8.encode(data, vp, [0] * 3)
    filler = torch.zeros(96 << 20, dtype=torch.float32, device="cuda:0")
    side = torch.cuda.Stream()
    for trial in range(6):
        query = rng.random(dim, dtype=np.float32)
        staged = torch.from_numpy(query).cuda()
        dq = torch.full((dim,), float("nan"), device="cuda:0")
        torch.cuda.synchronize()
        stream = side if trial % 2 else torch.cuda.current_stream()
        with torch.cuda.stream(stream):
            for _ in range(6):
                filler.add_(1.0)  # ~ms of queued work in front of the real values
            dq.copy_(staged)
            qs = sh.encode_query(dq)  # stream=None -> torch's current stream on the tensor's device
        ids, sc = sh.topk(qs, 20)
        wi, ws = one.topk(one.encode_query(query), 20)
        assert np.array_equal(ids, wi), trial
        assert_bits_equal(sc, ws, f"trial {trial}")
    # a store encoded from a tensor that is still being written
    src = torch.from_numpy(data).cuda()
    dst = torch.zeros_like(src)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(6):
            filler.add_(1.0)
        dst.copy_(src)
        sh2 = qa.ShardedVectorsU8.encode(dst, vp, [0, 0])
    got = np.concatenate([sh2.shard(g)[0].storage_bytes() for g in range(2)])
    assert np.array_equal(got, one.storage_bytes())
    torch.cuda.synchronize()


def test_shard_view_keeps_its_sharded_store_alive():
    import gc

    rng = np.random.default_rng(22)
    data = rng.random((4000, 32), dtype=np.float32)
    vp = qa.VectorParameters(32, 4000, D.Dot, False)
    view, base = qa.ShardedVectorsU8.encode(data, vp, [0, 0]).shard(1)  # the parent is dropped right here
    gc.collect()
    one = qa.EncodedVectorsU8.encode(data, vp)
    assert base == 2000
    assert np.array_equal(view.storage_bytes(), one.storage_bytes()[2000:])
    q = rng.random(32, dtype=np.float32)
    assert_bits_equal(view.score_all(view.encode_query(q)), one.score_all(one.encode_query(q))[2000:], "view after parent drop")


def test_sharded_topk_at_the_documented_limits():
    """include/quantization_amd.h: k <= 1024 and shards x k <= 8192 merge slots.  AT the limits (8 shards x k = 1024, and
    4 shards x 1024 in a batch) the answer equals the single handle's; one past either limit is an argument error, not a
    wrong answer or a fault."""
    rng = np.random.default_rng(77)
    n, dim = 40_000, 32
    data = rng.random((n, dim), dtype=np.float32)
    data[1000:1400] = data[0]  # a tie group of 401 rows across the shard boundaries
    vp = qa.VectorParameters(dim, n, D.Dot, False)
    one = qa.EncodedVectorsU8.encode(data, vp)
    q = rng.random(dim, dtype=np.float32)
    q1 = one.encode_query(q)
    sh8 = qa.ShardedVectorsU8.encode(data, vp, [0] * 8)
    for largest in (True, False):
        _same_topk(sh8.topk(sh8.encode_query(q), 1024, largest=largest), one.topk(q1, 1024, largest=largest))
    with pytest.raises(qa.EncodingError):  # k > 1024
        sh8.topk(sh8.encode_query(q), 1025)
    with pytest.raises(qa.EncodingError):
        one.topk(q1, 1025)
    sh9 = qa.ShardedVectorsU8.encode(data, vp, [0] * 9)  # 9 x 1024 > 8192 merge slots
    with pytest.raises(qa.EncodingError, match="8192"):
        sh9.topk(sh9.encode_query(q), 1024)
    _same_topk(sh9.topk(sh9.encode_query(q), 910), one.topk(q1, 910))  # 9 x 910 = 8190 fits
    # batched: 3 queries x k = 1024 over 4 shards
    queries = rng.random((3, dim), dtype=np.float32)
    sh4 = qa.ShardedVectorsU8.encode(data, vp, [0] * 4)
    ids, sc = sh4.topk_batch(sh4.encode_query_batch(queries), 1024)
    ids1, sc1 = one.topk_batch(one.encode_query_batch(queries), 1024)
    assert np.array_equal(ids, ids1)
    assert_bits_equal(sc, sc1, "batched top-1024 over 4 shards")


def test_peer_access_outcome_is_recorded():
    """Construction records, per shard, how its device reaches devices[0] (never ignored): logical shards of one GPU are
    'same device'; a second real GPU is 'enabled' or carries the reason copies are staged."""
    rng = np.random.default_rng(3)
    data = rng.random((1000, 16), dtype=np.float32)
    vp = qa.VectorParameters(16, 1000, D.Dot, False)
    sh = qa.ShardedVectorsU8.encode(data, vp, [0, 0, 0])
    assert [sh.peer_access(g)[0] for g in range(3)] == ["same device"] * 3
    if qa.lib().qamd_device_count() >= 2:
        sh2 = qa.ShardedVectorsU8.encode(data, vp, [0, 1])
        state, why = sh2.peer_access(1)
        assert state in ("enabled", "unavailable", "failed") and why
    with pytest.raises(qa.EncodingError):
        sh.peer_access(3)
