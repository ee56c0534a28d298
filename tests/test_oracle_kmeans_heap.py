"""Pins the oracle's restatement of the reference's k-means (quantization/src/kmeans.rs:7-167,
encoded_vectors_pq.rs:278-342) and of the caller's 30-entry heap
(demos/src/ann_benchmark_data.rs:20-33,151-167) against independent numpy restatements of the same
source lines.  CPU only."""
import numpy as np
import pytest


def numpy_kmeans(data, max_threads, max_iterations=100, accuracy=np.float32(1e-5)):
    """kmeans.rs line by line in numpy (f32 / f64 as the Rust types), no empty-cluster handling."""
    n, dim = data.shape
    cen = data[:256].copy()                                                       # :25
    trace = []
    for _ in range(max_iterations):
        idx = np.zeros(n, dtype=np.uint32)
        for i in range(n):                                                        # update_indexes :139-166
            best, best_i = np.float32(3.40282347e+38), 0
            for k in range(256):
                d = np.float32(0)
                for j in range(dim):
                    t = np.float32(data[i, j] - cen[k, j])
                    d = np.float32(d + np.float32(t * t))
                if d < best:
                    best, best_i = d, k
            idx[i] = best_i
        trace.append(idx)
        acc = np.zeros((256, dim), dtype=np.float64)
        cnt = np.zeros(256, dtype=np.int64)
        per = n // max_threads                                                    # :77
        for w in range(max_threads):
            lo, hi = per * w, (n if w + 1 == max_threads else per * (w + 1))
            part = np.zeros((256, dim), dtype=np.float64)
            for r in range(lo, hi):
                cnt[idx[r]] += 1
                part[idx[r]] += data[r].astype(np.float64)                        # :90-92, row order
            acc += part                                                           # :101-107
        assert (cnt > 0).all(), "test data must not produce an empty cluster"
        acc /= cnt[:, None].astype(np.float64)
        new = acc.astype(np.float32)
        diff = np.float32(0)
        for v in np.abs(cen - new).reshape(-1):                                   # :125-135 sequential f32
            diff = np.float32(diff + v)
        cen = new
        if diff < accuracy:
            break
    return cen, trace


@pytest.mark.parametrize("threads", [1, 3])
def test_kmeans_matches_independent_restatement(qo, threads):
    rng = np.random.default_rng(5)
    # 256 tight, well separated blobs: no empty cluster, a handful of iterations
    centers = rng.random((256, 2)).astype(np.float32) * 100
    pts = np.repeat(centers, 2, axis=0) + rng.normal(0, 0.01, (512, 2)).astype(np.float32)
    data = np.ascontiguousarray(pts[rng.permutation(512)], dtype=np.float32)
    want, trace = numpy_kmeans(data, threads)
    cen, its, empties, got_trace = qo.kmeans(data, max_threads=threads, trace=True)
    assert empties == 0 and its == len(trace)
    for a, b in zip(got_trace, trace):
        assert np.array_equal(a, b)
    assert np.array_equal(cen.view(np.uint32), want.view(np.uint32))


def test_kmeans_worker_partition_changes_only_the_f64_order(qo):
    rng = np.random.default_rng(6)
    data = rng.random((4000, 4), dtype=np.float32)
    c1, it1, e1 = qo.kmeans(data, max_threads=1)
    c7, it7, e7 = qo.kmeans(data, max_threads=7)
    assert e1 == 0 and e7 == 0
    # same algorithm, different f64 partial-sum order: equal to within an f32 ulp or two
    assert np.allclose(c1, c7, rtol=1e-6, atol=1e-7)


def test_find_centroids_layout_and_small_counts(qo):
    rng = np.random.default_rng(7)
    data = rng.random((200, 10), dtype=np.float32)
    cen, its, em = qo.find_centroids(data, 4)                # count <= 256: the vectors themselves (:290-297)
    assert np.array_equal(cen[:200], data) and not cen[200:].any()
    data = rng.random((1500, 10), dtype=np.float32)
    rows = qo.pq_sample_rows(1500)
    assert rows.size == 1500 and np.array_equal(rows, np.arange(1500))
    cen, its, em = qo.find_centroids(data, 4, max_threads=2)  # chunks 4, 4, 2 (last one short, :116-121)
    assert its.size == 3 and (its >= 1).all()
    for c, (lo, hi) in enumerate([(0, 4), (4, 8), (8, 10)]):
        sub, it, e = qo.kmeans(np.ascontiguousarray(data[:, lo:hi]), max_threads=2, chunk_index=c)
        assert it == its[c]
        assert np.array_equal(cen[:, lo:hi], sub)             # written at the chunk's columns (:336-338)


def test_heap_keeps_the_k_smallest_first_come_on_ties(qo):
    rng = np.random.default_rng(8)
    sc = rng.random(20000).astype(np.float32)
    ids, got = qo.topk_heap(sc, 30)
    order = np.argsort(sc, kind="stable")[:30]
    assert np.array_equal(ids, order) and np.array_equal(got, sc[order])
    # heavy ties: the score multiset is the 30 smallest; every strictly-better row is present; rows
    # of the boundary value are early rows of that value (a later equal score never displaces, :157)
    sc = rng.integers(0, 7, 5000).astype(np.float32)
    ids, got = qo.topk_heap(sc, 30)
    assert np.array_equal(np.sort(got), np.sort(sc)[:30]) and np.all(np.diff(got) >= 0)
    assert np.array_equal(sc[ids], got)
    boundary = got[-1]
    better = np.flatnonzero(sc < boundary)
    assert set(better) <= set(ids.tolist())
    n_boundary = 30 - better.size
    first_boundary_rows = np.flatnonzero(sc == boundary)[:n_boundary]
    assert set(ids.tolist()) == set(better) | set(first_boundary_rows)
    # fewer rows than k
    ids, got = qo.topk_heap(np.array([3, 1, 2], dtype=np.float32), 30)
    assert ids.tolist() == [1, 2, 0] and got.tolist() == [1, 2, 3]
