"""Streaming encoders (`qamd_*_encoder_begin / observe / push / finish`): the reference's contract is a
clonable iterator walked twice in bounded memory (encoded_vectors_u8.rs:34-40,57,73-118;
encoded_storage.rs:17-25).  The result must be byte-identical to the one-shot call and to the
oracle; stop_condition is polled per batch."""
import numpy as np
import pytest

from util import assert_bits_equal

pytestmark = pytest.mark.gpu

qa = pytest.importorskip("quantization_amd")
torch = pytest.importorskip("torch")
D = qa.DistanceType


def batches_of(data, size):
    return lambda: (data[i:i + size] for i in range(0, data.shape[0], size))


def same_u8(a, b):
    ma, mb = a.metadata, b.metadata
    for key in ("alpha", "offset", "multiplier"):
        assert np.float32(ma[key]).view(np.uint32) == np.float32(mb[key]).view(np.uint32), key
    assert ma["actual_dim"] == mb["actual_dim"]


@pytest.mark.parametrize("dist,invert", [(D.Dot, False), (D.L2, True), (D.L1, False)])
@pytest.mark.parametrize("dim", [65, 768])
def test_u8_stream_equals_one_shot_and_oracle(dist, invert, dim, qo):
    rng = np.random.default_rng(dim)
    n = 20_011
    data = (rng.random((n, dim), dtype=np.float32) * 2 - 1).astype(np.float32)
    vp = qa.VectorParameters(dim, n, dist, invert)
    one = qa.EncodedVectorsU8.encode(data, vp)
    st = qa.EncodedVectorsU8.encode_stream(batches_of(data, 1777), vp)  # ragged host batches
    same_u8(one, st)
    rows, meta = qo.u8_encode(data, int(dist), invert)
    assert np.array_equal(st.storage_bytes(), rows)
    assert np.array_equal(one.storage_bytes(), rows)
    assert np.float32(meta.alpha).view(np.uint32) == np.float32(st.metadata["alpha"]).view(np.uint32)
    # quantile branch (count <= 100 000: the sample is every vector, deterministic in the reference)
    oneq = qa.EncodedVectorsU8.encode(data, vp, quantile=0.98)
    stq = qa.EncodedVectorsU8.encode_stream(batches_of(data, 4096), vp, quantile=0.98)
    same_u8(oneq, stq)
    rows_q, _ = qo.u8_encode(data, int(dist), invert, quantile=0.98)
    assert np.array_equal(stq.storage_bytes(), rows_q)


def test_u8_stream_2m_x_768_in_64k_batches(qo):
    """2M x 768 in 64k-row batches == the one-shot call == the oracle (sampled rows)."""
    n, dim, bs = 2_000_000, 768, 65536
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    data = torch.rand((n, dim), generator=g, device=dev)
    vp = qa.VectorParameters(dim, n, D.Dot, False)
    polls = []
    st = qa.EncodedVectorsU8.encode_stream(batches_of(data, bs), vp, stop_condition=lambda: polls.append(1) and False)
    assert len(polls) >= 2 * ((n + bs - 1) // bs), "stop_condition is polled per batch in both passes"
    one = qa.EncodedVectorsU8.encode(data, vp)
    same_u8(one, st)
    a = st.storage_bytes(out=torch.empty(n * 772, dtype=torch.uint8, device=dev))
    b = one.storage_bytes(out=torch.empty(n * 772, dtype=torch.uint8, device=dev))
    assert torch.equal(a, b)
    md = st.metadata
    ids = torch.randint(0, n, (3000,), generator=g, device=dev)
    o_rows, _ = qo.u8_encode_with(data[ids].cpu().numpy(), qo.DOT, False, float(md["alpha"]), float(md["offset"]))
    assert np.array_equal(a.view(n, 772)[ids].cpu().numpy(), o_rows)
    # > 100 000 rows with a quantile: streaming and one-shot pick the same evenly strided sample
    stq = qa.EncodedVectorsU8.encode_stream(batches_of(data[:300_000], bs), qa.VectorParameters(dim, 300_000, D.Dot, False),
                                            quantile=0.99)
    oneq = qa.EncodedVectorsU8.encode(data[:300_000], qa.VectorParameters(dim, 300_000, D.Dot, False), quantile=0.99)
    same_u8(stq, oneq)


def test_u8_stream_stop_and_count_errors():
    rng = np.random.default_rng(3)
    data = rng.random((5000, 32), dtype=np.float32)
    vp = qa.VectorParameters(32, 5000, D.Dot, False)
    calls = []

    def stop():
        calls.append(1)
        return len(calls) > 7  # flips in the middle of the second pass

    with pytest.raises(qa.EncodingError) as e:
        qa.EncodedVectorsU8.encode_stream(batches_of(data, 1000), vp, stop_condition=stop)
    assert e.value.stopped
    with pytest.raises(qa.EncodingError) as e:  # the iterator yields fewer rows than vp.count
        qa.EncodedVectorsU8.encode_stream(batches_of(data[:4000], 1000), vp)
    assert e.value.kind == "ArgumentsError" and "count" in str(e.value)
    with pytest.raises(qa.EncodingError):       # ... or more
        qa.EncodedVectorsU8.encode_stream(batches_of(data, 1000), qa.VectorParameters(32, 4500, D.Dot, False))
    # empty store (tests/empty_storage.rs)
    z = qa.EncodedVectorsU8.encode_stream(lambda: iter(()), qa.VectorParameters(32, 0, D.Dot, False))
    assert z.count == 0 and z.metadata["alpha"] == 0


@pytest.mark.parametrize("dim,store", [(1024, qa.BitsStoreType.U8), (387, qa.BitsStoreType.U128), (33, qa.BitsStoreType.U8)])
def test_binary_stream_equals_one_shot_and_oracle(dim, store, qo):
    rng = np.random.default_rng(dim)
    n = 30_001
    data = rng.standard_normal((n, dim)).astype(np.float32)
    vp = qa.VectorParameters(dim, n, D.Dot, False)
    one = qa.EncodedVectorsBin.encode(data, vp, store=store)
    st = qa.EncodedVectorsBin.encode_stream(batches_of(data, 3333), vp, store=store)
    want = qo.bin_encode(data, store=int(store))
    assert np.array_equal(st.storage_bytes(), want)
    assert np.array_equal(one.storage_bytes(), want)
    dd = torch.from_numpy(data).cuda()
    st2 = qa.EncodedVectorsBin.encode_stream(batches_of(dd, 4096), vp, store=store)  # device batches
    assert np.array_equal(st2.storage_bytes(), want)
    calls = []
    with pytest.raises(qa.EncodingError) as e:
        qa.EncodedVectorsBin.encode_stream(batches_of(data, 3333), vp, store=store,
                                           stop_condition=lambda: calls.append(1) or len(calls) > 3)
    assert e.value.stopped


def test_pq_stream_equals_one_shot_and_oracle(qo):
    rng = np.random.default_rng(21)
    n, dim, chunk = 25_000, 48, 4
    data = rng.random((n, dim), dtype=np.float32)
    cen = rng.random((256, dim), dtype=np.float32)
    vp = qa.VectorParameters(dim, n, D.L2, False)
    # given centroids: one walk, codes == oracle
    st = qa.EncodedVectorsPQ.encode_stream(batches_of(data, 2048), vp, chunk, centroids=cen)
    assert np.array_equal(st.storage_bytes(), qo.pq_encode(data, chunk, cen))
    # trained: two walks, byte-identical to the one-shot call (same strided sample, same k-means)
    one = qa.EncodedVectorsPQ.encode(data, vp, chunk, max_kmeans_threads=3)
    st2 = qa.EncodedVectorsPQ.encode_stream(batches_of(data, 2999), vp, chunk, max_kmeans_threads=3)
    assert np.array_equal(st2.centroids.view(np.uint32), one.centroids.view(np.uint32))
    assert np.array_equal(st2.storage_bytes(), one.storage_bytes())
    # count <= 256: the vectors themselves (:290-297)
    small = data[:100]
    vps = qa.VectorParameters(dim, 100, D.Dot, False)
    st3 = qa.EncodedVectorsPQ.encode_stream(batches_of(small, 33), vps, chunk)
    assert np.array_equal(st3.centroids, qo.pq_centroids_small(small))
    assert np.array_equal(st3.storage_bytes(), qo.pq_encode(small, chunk, st3.centroids))
    z = qa.EncodedVectorsPQ.encode_stream(lambda: iter(()), qa.VectorParameters(dim, 0, D.Dot, False), chunk)
    assert z.count == 0


@pytest.mark.parametrize("dim,chunk", [(640, 4), (576, 3), (192, 4)])
def test_pq_stream_builds_the_scan_image_batch_by_batch(dim, chunk, qo):
    """Rows of several LUT slices (m = 160, 192: the planar scan image) and of 48 chunks (two rows per ring row) encoded by the
    streaming encoder in ragged host and device batches, by the one-shot encoder and loaded from reference-format rows: the
    same row bytes, and the whole-store scan (which reads the scan image) gives the oracle's score_point_sse bits."""
    rng = np.random.default_rng(dim)
    n = 21_001
    m = dim // chunk
    data = rng.random((n, dim), dtype=np.float32)
    cen = (rng.random((256, dim), dtype=np.float32) - 0.5).astype(np.float32)
    vp = qa.VectorParameters(dim, n, D.Dot, True)
    rows = qo.pq_encode(data, chunk, cen)
    query = (rng.random(dim, dtype=np.float32) - 0.5).astype(np.float32)
    want = qo.pq_score_all(rows, qo.pq_encode_query(query, chunk, cen, qo.DOT, True), order=qo.ORDER_SSE)
    stores = [qa.EncodedVectorsPQ.encode_stream(batches_of(data, 2777), vp, chunk, centroids=cen),
              qa.EncodedVectorsPQ.encode_stream(batches_of(torch.from_numpy(data).cuda(), 4096), vp, chunk, centroids=cen),
              qa.EncodedVectorsPQ.encode(data, vp, chunk, centroids=cen),
              qa.EncodedVectorsPQ.from_storage(rows, vp, chunk, cen)]
    for enc in stores:
        assert enc.scan_kernel()[0].startswith("pq_scan_skew_kernel") and enc.scan_kernel()[1] == (2 if m > 128 else 1)
        assert np.array_equal(enc.storage_bytes(), rows)
        assert_bits_equal(enc.score_all(enc.encode_query(query)), want, f"scan of the streamed store, m={m}")


def test_aborted_and_failed_encoders_give_their_memory_back():
    """An encoder that is aborted (stop_condition, an exception in the caller's iterator, a count
    mismatch at finish) frees the store it was building."""
    rng = np.random.default_rng(9)
    n, dim = 400_000, 256  # ~100 MB of codes per store
    data = rng.random((20_000, dim), dtype=np.float32)
    vp = qa.VectorParameters(dim, n, D.Dot, False)

    def free_now():
        torch.cuda.synchronize()
        return torch.cuda.mem_get_info(0)[0]

    def bad_iter():
        yield data
        raise RuntimeError("the caller's iterator failed")

    base = free_now()
    for _ in range(5):
        with pytest.raises(RuntimeError, match="iterator failed"):
            qa.EncodedVectorsU8.encode_stream(bad_iter, vp, alpha_offset=(0.01, 0.0))
        with pytest.raises(qa.EncodingError):  # 20 000 of 400 000 rows pushed
            qa.EncodedVectorsU8.encode_stream(lambda: iter([data]), vp, alpha_offset=(0.01, 0.0))
        with pytest.raises(qa.EncodingError):
            qa.EncodedVectorsBin.encode_stream(lambda: iter([data]), vp)
        with pytest.raises(qa.EncodingError):
            qa.EncodedVectorsPQ.encode_stream(lambda: iter([data]), vp, 8, centroids=rng.random((256, dim), dtype=np.float32))
    assert abs(free_now() - base) < (32 << 20), "aborted encoders leaked device memory"
