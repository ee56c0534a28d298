"""GPU k-means (csrc/pq.hip km_* kernels) against the oracle's restatement of
quantization/src/kmeans.rs:7-167 + encoded_vectors_pq.rs:278-342.

The reference's two random draws are fixed: the sample is the evenly strided rows
floor(k * count / S) on both sides, and the data produce no empty cluster (asserted through
kmeans_info / the oracle's counter).  Everything else is the reference's arithmetic in the
reference's order — assignment, f64 sums split over `max_kmeans_threads` contiguous row ranges and
merged in worker order, f32 shift sum, stopping rule — so the centroids must be BIT-identical, and
with them every code."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

qa = pytest.importorskip("quantization_amd")
D = qa.DistanceType


@pytest.mark.parametrize("threads", [1, 3, 8])
@pytest.mark.parametrize("dim,chunk", [(32, 4), (30, 8), (16, 1)])
def test_gpu_kmeans_bit_equals_oracle(threads, dim, chunk, qo):
    rng = np.random.default_rng(dim * 100 + chunk)
    n = 6000
    data = rng.random((n, dim), dtype=np.float32)
    vp = qa.VectorParameters(dim, n, D.Dot, False)
    enc = qa.EncodedVectorsPQ.encode(data, vp, chunk, max_kmeans_threads=threads)
    iters, empties = enc.kmeans_info()
    want, its, o_empties = qo.find_centroids(data, chunk, qo.pq_sample_rows(n), max_threads=threads)
    assert o_empties == 0 and empties == 0, "parity is conditional on no empty cluster"
    assert iters == int(its.max())
    assert np.array_equal(enc.centroids.view(np.uint32), want.view(np.uint32)), "centroids differ from kmeans.rs"
    assert np.array_equal(enc.storage_bytes(), qo.pq_encode(data, chunk, want))


def test_gpu_kmeans_strided_sample_above_10k_rows(qo):
    rng = np.random.default_rng(77)
    n, dim, chunk = 40_000, 24, 6
    data = rng.random((n, dim), dtype=np.float32)
    enc = qa.EncodedVectorsPQ.encode(data, qa.VectorParameters(dim, n, D.L2, False), chunk, max_kmeans_threads=2)
    rows = qo.pq_sample_rows(n)
    assert rows.size == 10_000 and rows[1] == 4 and rows[-1] == 39_996
    want, its, em = qo.find_centroids(data, chunk, rows, max_threads=2)
    assert em == 0 and enc.kmeans_info()[1] == 0
    assert np.array_equal(enc.centroids.view(np.uint32), want.view(np.uint32))


def test_gpu_kmeans_empty_cluster_rule_is_the_stated_one(qo):
    """Duplicated rows make clusters empty from the first iteration: the reference re-seeds from
    thread_rng (kmeans.rs:111-118, unreproducible); product and oracle share the stated hash rule,
    so they still agree — and report that the conditional parity precondition does NOT hold."""
    rng = np.random.default_rng(78)
    base = rng.random((300, 8), dtype=np.float32)
    data = np.ascontiguousarray(np.concatenate([base[:100]] * 3 + [base]), dtype=np.float32)  # first 256 rows repeat
    n = data.shape[0]
    enc = qa.EncodedVectorsPQ.encode(data, qa.VectorParameters(8, n, D.Dot, False), 4)
    want, its, em = qo.find_centroids(data, 4, qo.pq_sample_rows(n))
    assert em > 0 and enc.kmeans_info()[1] == em
    assert np.array_equal(enc.centroids.view(np.uint32), want.view(np.uint32))
