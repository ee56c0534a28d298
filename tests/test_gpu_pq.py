"""HIP PQ path vs the oracle: codes, LUT and scores bit-exact GIVEN centroids (the reference's
k-means is randomised: centroid values are parity-unpinned); the reference's
tolerance spec (quantization/tests/test_pq.rs) is re-run on trained centroids."""
import numpy as np
import pytest

from util import assert_bits_equal

pytestmark = pytest.mark.gpu

qa = pytest.importorskip("quantization_amd")
D = qa.DistanceType

CASES = [(513, 65, 1), (300, 64, 2), (1000, 768, 8), (200, 100, 7), (64, 16, 16), (150, 1024, 2), (90, 33, 4)]


def _data(n, dim, seed=42):
    rng = np.random.default_rng(seed)
    return rng.random((n, dim), dtype=np.float32), rng.random(dim, dtype=np.float32), \
        rng.random((256, dim), dtype=np.float32)


@pytest.mark.parametrize("n,dim,chunk", CASES)
@pytest.mark.parametrize("dist", [D.Dot, D.L1, D.L2])
@pytest.mark.parametrize("invert", [False, True])
def test_pq_bit_exact_given_centroids(qo, n, dim, chunk, dist, invert):
    data, query, cen = _data(n, dim)
    vp = qa.VectorParameters(dim, n, dist, invert)
    enc = qa.EncodedVectorsPQ.encode(data, vp, chunk, centroids=cen)
    assert qa.EncodedVectorsPQ.get_quantized_vector_size(vp, chunk) == qo.pq_chunks(dim, chunk)
    rows = qo.pq_encode(data, chunk, cen)
    assert np.array_equal(enc.storage_bytes(), rows), "PQ codes differ"
    q = enc.encode_query(query)
    lut = qo.pq_encode_query(query, chunk, cen, int(dist), invert)
    assert_bits_equal(q.lut, lut, "LUT")
    want = qo.pq_score_all(rows, lut, order=qo.ORDER_SSE)
    assert_bits_equal(enc.score_all(q), want, "score_all (score_point_sse order)")
    for i in (0, n // 2, n - 1):
        assert_bits_equal([enc.score_point(q, i)], [want[i]], "score_point")
        j = (i * 5 + 1) % n
        assert_bits_equal([enc.score_internal(i, j)],
                          [qo.pq_score_internal(rows, dim, chunk, cen, int(dist), invert, i, j)], "score_internal")
    ids = np.array([n - 1, 3, 3, 0], dtype=np.uint32)
    assert_bits_equal(enc.score_ids(q, ids), want[ids], "score_ids")


def test_pq_large_scan_uses_lds_lut(qo):
    """n >= 4096 takes the LDS-resident-LUT kernel (96 KiB at m = 96)."""
    n, dim, chunk = 20000, 768, 8
    data, query, cen = _data(n, dim, seed=7)
    enc = qa.EncodedVectorsPQ.encode(data, qa.VectorParameters(dim, n, D.Dot, False), chunk, centroids=cen)
    rows = qo.pq_encode(data, chunk, cen)
    assert np.array_equal(enc.storage_bytes(), rows)
    lut = qo.pq_encode_query(query, chunk, cen, qo.DOT, False)
    assert_bits_equal(enc.score_all(enc.encode_query(query)), qo.pq_score_all(rows, lut, order=qo.ORDER_SSE), "LDS LUT scan")


def test_pq_small_count_centroids_are_the_vectors(qo):
    """encoded_vectors_pq.rs:290-297: count <= 256."""
    n, dim, chunk = 100, 40, 4
    data, query, _ = _data(n, dim, seed=3)
    enc = qa.EncodedVectorsPQ.encode(data, qa.VectorParameters(dim, n, D.L2, False), chunk)
    cen = qo.pq_centroids_small(data)
    assert np.array_equal(enc.centroids, cen)
    rows = qo.pq_encode(data, chunk, cen)
    assert np.array_equal(enc.storage_bytes(), rows)
    # every vector is its own nearest centroid: distance-to-self scores 0 under L2
    for i in (0, 50, 99):
        assert enc.score_internal(i, i) == 0.0


@pytest.mark.parametrize("dist", [D.Dot, D.L1, D.L2])
@pytest.mark.parametrize("invert", [False, True])
def test_pq_reference_tolerance_spec_with_trained_centroids(qo, dist, invert):
    """quantization/tests/test_pq.rs:16-50: 513 x 65, chunk 1, |score - metric| < dim*0.05."""
    n, dim = 513, 65
    data, query, _ = _data(n, dim)
    enc = qa.EncodedVectorsPQ.encode(data, qa.VectorParameters(dim, n, dist, invert), 1, max_kmeans_threads=1)
    q = enc.encode_query(query)
    scores = enc.score_all(q)
    for i in range(n):
        orig = qo.metric_f32(int(dist), query, data[i])
        orig = -orig if invert else orig
        assert abs(scores[i] - orig) < dim * 0.05
    # and the whole pipeline is self-consistent with the oracle GIVEN those centroids
    cen = enc.centroids
    rows = qo.pq_encode(data, 1, cen)
    assert np.array_equal(enc.storage_bytes(), rows)
    lut = qo.pq_encode_query(query, 1, cen, int(dist), invert)
    assert_bits_equal(scores, qo.pq_score_all(rows, lut, order=qo.ORDER_SSE), "scores given trained centroids")
    for i in range(0, n - 1, 64):
        orig = qo.metric_f32(int(dist), data[i], data[i + 1])
        orig = -orig if invert else orig
        assert abs(enc.score_internal(i, i + 1) - orig) < dim * 0.05


def test_pq_kmeans_reduces_distortion():
    """k-means (kmeans.rs) quality: trained centroids beat the first-256-rows initialisation."""
    rng = np.random.default_rng(0)
    n, dim, chunk = 4000, 32, 4
    centers = rng.standard_normal((64, dim)).astype(np.float32) * 3
    data = (centers[rng.integers(0, 64, n)] + rng.standard_normal((n, dim)).astype(np.float32) * 0.3).astype(np.float32)
    vp = qa.VectorParameters(dim, n, D.L2, False)
    trained = qa.EncodedVectorsPQ.encode(data, vp, chunk)
    naive = qa.EncodedVectorsPQ.encode(data, vp, chunk, centroids=np.concatenate([data[:256]]))

    def distortion(enc):
        cen, codes = enc.centroids, enc.storage_bytes()
        rec = np.concatenate([cen[codes[:, c], c * chunk:(c + 1) * chunk] for c in range(dim // chunk)], axis=1)
        return float(((rec - data) ** 2).sum(axis=1).mean())

    assert distortion(trained) < distortion(naive) * 0.9


def test_pq_save_load_empty_and_stop(qo, tmp_path):
    import json
    n, dim, chunk = 300, 24, 5
    data, query, cen = _data(n, dim, seed=8)
    vp = qa.VectorParameters(dim, n, D.Dot, True)
    enc = qa.EncodedVectorsPQ.encode(data, vp, chunk, centroids=cen)
    enc.save(tmp_path / "p.bin", tmp_path / "p.json")
    js = json.load(open(tmp_path / "p.json"))
    assert list(js.keys()) == ["centroids", "vector_division", "vector_parameters"]
    assert js["vector_division"][-1] == {"start": 20, "end": 24} and len(js["vector_division"]) == 5
    assert np.array_equal(np.array(js["centroids"], dtype=np.float32), cen)
    back = qa.EncodedVectorsPQ.load(tmp_path / "p.bin", tmp_path / "p.json", vp)
    assert_bits_equal(back.score_all(back.encode_query(query)), enc.score_all(enc.encode_query(query)), "reload")
    # empty storage (quantization/tests/empty_storage.rs)
    vp0 = qa.VectorParameters(dim, 0, D.Dot, False)
    e0 = qa.EncodedVectorsPQ.encode(np.zeros((0, dim), np.float32), vp0, chunk)
    e0.save(tmp_path / "e.bin", tmp_path / "e.json")
    b0 = qa.EncodedVectorsPQ.load(tmp_path / "e.bin", tmp_path / "e.json", vp0)
    assert b0.score_all(b0.encode_query(query)).size == 0
    with pytest.raises(qa.EncodingError) as e:
        qa.EncodedVectorsPQ.encode(data, vp, chunk, centroids=cen, stop_condition=lambda: True)
    assert e.value.stopped


@pytest.mark.parametrize("dim,chunk", [(1024, 2), (65, 1), (200, 1), (100, 3), (768, 4), (130, 2), (4096, 8)])
def test_pq_sliced_fast_scan_any_m(qo, dim, chunk):
    """n >= 4096 takes the LDS fast kernel for every m >= 16: m % 16 != 0 (padded last piece,
    m % 4 tail) and LUTs larger than LDS scanned in slices (m = 512 is the reference bench's own
    configuration, demos/benches/pq.rs:12-46)."""
    n = 6000
    rng = np.random.default_rng(dim * 7 + chunk)
    m = qo.pq_chunks(dim, chunk)
    cen = rng.random((256, dim), dtype=np.float32)
    rows = rng.integers(0, 256, size=(n, m), dtype=np.uint8)
    query = rng.random(dim, dtype=np.float32)
    for dist, invert in ((D.Dot, False), (D.L2, True)):
        enc = qa.EncodedVectorsPQ.from_storage(rows, qa.VectorParameters(dim, n, dist, invert), chunk, cen)
        assert np.array_equal(enc.storage_bytes(), rows)
        lut = qo.pq_encode_query(query, chunk, cen, int(dist), invert)
        want = qo.pq_score_all(rows, lut, order=qo.ORDER_SSE)
        q = enc.encode_query(query)
        assert_bits_equal(enc.score_all(q), want, f"m={m}")
        ids = np.array([0, n - 1, 17], dtype=np.uint32)
        assert_bits_equal(enc.score_ids(q, ids), want[ids], "ids kernel")


@pytest.mark.parametrize("m,chunk,n", [(96, 8, 300_001), (96, 8, 4097), (64, 4, 70_003), (32, 2, 50_000), (96, 1, 6007),
                                       (128, 8, 200_003), (128, 2, 4100),
                                       # m = 48: two store rows per ring row (odd and even row counts, a lone last row)
                                       (48, 16, 300_001), (48, 2, 4096), (48, 1, 4097), (48, 4, 70_002), (48, 8, 33),
                                       # m = 16: two store rows per 32-chunk ring row, a row end every four steps (the eight
                                       # lags span two store rows)
                                       (16, 8, 300_001), (16, 1, 4096), (16, 4, 4097), (16, 2, 70_003), (16, 8, 4127),
                                       # m = 80 / 112: rows one 16-chunk piece short of a 96 / 128-chunk ring row; the missing chunks read
                                       # table columns of +0.0 (ragged row counts, the run / coalesced-store path, dim not a multiple of the chunk)
                                       (80, 8, 300_001), (80, 1, 4097), (80, 2, 70_003), (112, 2, 70_003), (112, 1, 4100), (112, 4, 33),
                                       (80, 1, 1_100_003), (112, 1, 530_001),
                                       # ... and every m % 4 == 0 below 128 the same way: rows on their pitch of whole 16-byte pieces, ring rows of
                                       # the next 32, zero table columns past m (120 = dim 960 at chunk 8; 100 -> 112 -> 128; 88 -> 96; 36 -> 48 -> 64)
                                       (120, 8, 300_001), (100, 2, 70_003), (88, 1, 4097), (72, 4, 50_001), (36, 2, 40_000), (52, 1, 4099),
                                       (20, 4, 70_001), (124, 1, 1_100_003), (120, 1, 530_001), (68, 1, 33),
                                       # rows of several LUT slices, scanned from the planar image (96 + 96, 128 + 32, 128 + 96,
                                       # 96 x 3, four of 128)
                                       (192, 4, 100_003), (160, 1, 5000), (224, 1, 9001), (288, 2, 30_001), (512, 2, 20_011),
                                       # stores large enough for RUNS of four consecutive blocks per wave and the coalesced
                                       # 256-byte score store (a last run of 1, 2 and 3 blocks; two rows per ring row keeps the
                                       # per-block store on runs)
                                       (96, 1, 1_100_003), (32, 1, 1_048_577), (64, 1, 1_050_030), (128, 1, 530_001),
                                       (48, 1, 2_100_001), (16, 1, 2_100_003), (192, 1, 1_050_011)])
def test_pq_skewed_scan_shapes(qo, m, chunk, n):
    """m = 32 / 64 / 96 / 128 whole-store scans take pq_scan_skew_kernel (transposed LUT, quads skewed in time, rows through a
    per-wave LDS ring): same bits as the oracle's score_point_sse order for row counts that are not multiples of 16, waves
    with one block and with many, zero and negative-zero table entries, and the same top-k as the scores."""
    dim = m * chunk
    rng = np.random.default_rng(m * 131 + n)
    cen = (rng.random((256, dim), dtype=np.float32) - 0.5).astype(np.float32)
    cen[rng.integers(0, 256, size=40)] = 0.0  # whole zero centroids: +0.0 / -0.0 entries (invert) in every chunk table
    rows = rng.integers(0, 256, size=(n, m), dtype=np.uint8)
    query = (rng.random(dim, dtype=np.float32) - 0.5).astype(np.float32)
    for dist, invert in ((D.Dot, False), (D.Dot, True), (D.L2, False)):
        enc = qa.EncodedVectorsPQ.from_storage(rows, qa.VectorParameters(dim, n, dist, invert), chunk, cen)
        lut = qo.pq_encode_query(query, chunk, cen, int(dist), invert)
        want = qo.pq_score_all(rows, lut, order=qo.ORDER_SSE)
        q = enc.encode_query(query)
        if n >= 4096:
            assert enc.scan_kernel() == (("pq_scan_skew_kernel", 1) if m <= 128 else
                                         ("pq_scan_skew_kernel<SLICED>", {192: 2, 160: 2, 224: 2, 288: 3, 512: 4}[m]))
        assert_bits_equal(enc.score_all(q), want, f"m={m} n={n}")
        assert np.array_equal(enc.storage_bytes(), rows), "the row-major image is what export returns"
        pick = rng.integers(0, n, size=64).astype(np.uint32)
        assert_bits_equal(enc.score_ids(q, pick), want[pick], "random access reads the row-major image")
        for largest in (True, False):
            ids, sc = enc.topk(q, 30, largest=largest)
            order = np.lexsort((np.arange(n), -want if largest else want))[:30]
            assert_bits_equal(sc, want[order], "top-k scores")
            assert np.array_equal(np.sort(want[ids]), np.sort(want[order]))


def test_pq_skewed_and_older_scan_kernels_give_the_same_bits():
    """The same stores scanned by pq_scan_skew_kernel (default) and by pq_scan_fast_kernel (QAMD_PQ_SKEW=0, a developer switch
    that only the tools/lib build reads - the product library ignores it): identical score bits and identical top-k for whole
    rows (m = 16, 48, 64, 80, 96, 112, 128) and sliced rows (m = 192, 288)."""
    import hashlib
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, hashlib
sys.path.insert(0, %r)
import numpy as np
import quantization_amd as qa
D = qa.DistanceType
h = hashlib.sha256()
names = []
for m, chunk, n in ((16, 8, 60001), (48, 4, 40001), (64, 2, 9000), (80, 2, 45001), (96, 8, 50001), (112, 1, 33333), (128, 4, 30007), (192, 4, 20011), (288, 1, 7001)):
    rng = np.random.default_rng(m)
    dim = m * chunk
    cen = (rng.random((256, dim), dtype=np.float32) - 0.5).astype(np.float32)
    rows = rng.integers(0, 256, size=(n, m), dtype=np.uint8)
    enc = qa.EncodedVectorsPQ.from_storage(rows, qa.VectorParameters(dim, n, D.L2, True), chunk, cen)
    q = enc.encode_query((rng.random(dim, dtype=np.float32) - 0.5).astype(np.float32))
    names.append(enc.scan_kernel()[0])
    h.update(np.asarray(enc.score_all(q)).tobytes())
    ids, sc = enc.topk(q, 50, largest=False)
    h.update(np.asarray(ids).tobytes()); h.update(np.asarray(sc).tobytes())
print("DIGEST", h.hexdigest())
print("KERNELS", " ".join(sorted(set(names))))
""" % root
    dev_lib = os.path.join(root, "tools", "lib", "libquantization_amd_dev.so")
    if not os.path.exists(dev_lib):
        pytest.skip("tools/lib/libquantization_amd_dev.so not built (make -C quantization_amd/csrc dev)")
    digests, kernels = [], []
    # product library (the switch set, and ignored), developer library with the switch on and off
    for lib, skew in ((None, "0"), (dev_lib, "1"), (dev_lib, "0")):
        env = dict(os.environ, QAMD_PQ_SKEW=skew)
        env.pop("QAMD_LIB_PATH", None)
        if lib:
            env["QAMD_LIB_PATH"] = lib
        res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
        assert res.returncode == 0, res.stderr[-2000:]
        digests.append([ln for ln in res.stdout.splitlines() if ln.startswith("DIGEST")][-1])
        kernels.append([ln for ln in res.stdout.splitlines() if ln.startswith("KERNELS")][-1])
    assert digests[0] == digests[1] == digests[2]
    assert "fast" not in kernels[0], "the product library must ignore developer switches"
    assert "fast" not in kernels[1] and "skew" not in kernels[2], kernels
