"""BASELINE.json's full sizes on one MI355X, checked through size-independent properties
(exact integer linearity and checksums, complement symmetry, kernel-vs-kernel agreement) plus
oracle parity on sampled rows.  torch is plumbing: synthetic data in HBM and int64 checksums."""
import numpy as np
import pytest

from util import assert_bits_equal

pytestmark = pytest.mark.gpu

qa = pytest.importorskip("quantization_amd")
torch = pytest.importorskip("torch")
D = qa.DistanceType


def _free():
    import gc
    gc.collect()
    torch.cuda.empty_cache()


def test_c1_u8_dot_l2_10m_x_768(qo):
    """configs[1]: 10M x 768 scalar u8, dot + L2."""
    n, dim = 10_000_000, 768
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(42)
    # integer-valued data with alpha = 1, offset = 0: codes == values, multiplier == 1,
    # every offset == 0, so Dot scores are the exact integer dot products (< 2^24).
    data = torch.randint(0, 128, (n, dim), generator=g, device=dev, dtype=torch.int32).to(torch.float32)
    enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, D.Dot, False), alpha_offset=(1.0, 0.0))
    q1 = torch.randint(0, 64, (dim,), generator=g, device=dev).to(torch.float32)
    q2 = torch.randint(0, 64, (dim,), generator=g, device=dev).to(torch.float32)
    s1 = enc.score_all(enc.encode_query(q1), out=torch.empty(n, device=dev))
    s2 = enc.score_all(enc.encode_query(q2), out=torch.empty(n, device=dev))
    s12 = enc.score_all(enc.encode_query(q1 + q2), out=torch.empty(n, device=dev))
    torch.cuda.synchronize()
    assert torch.equal(s12, s1 + s2), "linearity in the query (exact integers)"
    colsum = torch.zeros(dim, dtype=torch.float64, device=dev)
    for lo in range(0, n, 1_000_000):  # checksum of checksums: sum_i dot_i == q . column sums
        colsum += data[lo:lo + 1_000_000].sum(dim=0, dtype=torch.float64)
    assert float(s1.sum(dtype=torch.float64)) == float((colsum * q1.double()).sum())
    # sampled rows against the oracle (bit-exact), and scan kernel vs random-access kernel
    ids = torch.randint(0, n, (4096,), generator=g, device=dev)
    sample = data[ids].cpu().numpy()
    rows, meta = qo.u8_encode_with(sample, qo.DOT, False, 1.0, 0.0)
    codes, qoff = qo.u8_encode_query(meta, q1.cpu().numpy())
    assert_bits_equal(s1[ids].cpu().numpy(), qo.u8_score_all(meta, rows, codes, qoff, order=qo.ORDER_AVX2), "sampled rows")
    qobj = enc.encode_query(q1)
    assert torch.equal(enc.score_ids(qobj, ids.to(torch.int32), out=torch.empty(4096, device=dev)), s1[ids])
    ti, ts = enc.topk(qobj, 30)
    best = torch.topk(s1, 30)
    assert np.array_equal(np.sort(ts)[::-1], best.values.cpu().numpy())
    del enc, s2, s12
    _free()

    # L2 on real-valued data: sampled oracle parity + agreement of the two kernels
    data = torch.rand((n, dim), generator=g, device=dev)
    enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, D.L2, False))
    md = enc.metadata
    q = torch.rand(dim, generator=g, device=dev)
    qobj = enc.encode_query(q)
    s = enc.score_all(qobj, out=torch.empty(n, device=dev))
    sample = data[ids].cpu().numpy()
    rows, meta = qo.u8_encode_with(sample, qo.L2, False, float(md["alpha"]), float(md["offset"]))
    codes, qoff = qo.u8_encode_query(meta, q.cpu().numpy())
    torch.cuda.synchronize()
    assert_bits_equal(s[ids].cpu().numpy(), qo.u8_score_all(meta, rows, codes, qoff, order=qo.ORDER_AVX2), "L2 sampled rows")
    assert torch.equal(enc.score_ids(qobj, ids.to(torch.int32), out=torch.empty(4096, device=dev)), s[ids])
    # the encode found the true global min/max
    assert float(md["offset"]) == float(data.min()) and np.float32(md["alpha"]) == np.float32(
        (np.float32(float(data.max())) - np.float32(float(data.min()))) / np.float32(127.0))
    del enc, data
    _free()


def test_c2_pq_10m_x_768_m96(qo):
    """configs[2]: 10M x 768, PQ m = 96 (chunk 8), ks = 256, LDS-resident LUT."""
    n, dim, chunk = 10_000_000, 768, 8
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    cen = np.random.default_rng(7).random((256, dim), dtype=np.float32)
    rows = torch.randint(0, 256, (n, 96), generator=g, device=dev, dtype=torch.uint8)
    vp = qa.VectorParameters(dim, n, D.Dot, False)
    enc = qa.EncodedVectorsPQ.from_storage(rows, vp, chunk, cen)
    query = np.random.default_rng(8).random(dim, dtype=np.float32)
    q = enc.encode_query(query)
    lut = qo.pq_encode_query(query, chunk, cen, qo.DOT, False)
    assert_bits_equal(q.lut, lut, "LUT")
    s = enc.score_all(q, out=torch.empty(n, device=dev))
    ids = torch.randint(0, n, (8192,), generator=g, device=dev)
    torch.cuda.synchronize()
    want = qo.pq_score_all(rows[ids].cpu().numpy(), lut, order=qo.ORDER_SSE)
    assert_bits_equal(s[ids].cpu().numpy(), want, "sampled rows vs oracle (SSE order)")
    # LDS fast kernel vs the random-access kernel on a permutation of every row
    perm = torch.randperm(n, generator=g, device=dev).to(torch.int32)
    sp = enc.score_ids(q, perm, out=torch.empty(n, device=dev))
    assert torch.equal(sp, s[perm.long()]), "scan kernel and ids kernel disagree"
    # encode side at a bounded size: GPU codes == oracle codes for real vectors
    data = np.random.default_rng(9).random((20000, dim), dtype=np.float32)
    e2 = qa.EncodedVectorsPQ.encode(data, qa.VectorParameters(dim, 20000, D.Dot, False), chunk, centroids=cen)
    assert np.array_equal(e2.storage_bytes()[:2000], qo.pq_encode(data[:2000], chunk, cen))
    del enc, rows
    _free()


def test_c3_binary_50m_x_1024(qo):
    """configs[3]: 50M x 1024 binary / Hamming (one GPU holds all 6.4 GB)."""
    n, dim = 50_000_000, 1024
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    rows = torch.randint(0, 256, (n, 128), generator=g, device=dev, dtype=torch.uint8)
    vp = qa.VectorParameters(dim, n, D.Dot, False)
    enc = qa.EncodedVectorsBin.from_storage(rows, vp)
    qv = torch.randn(dim, generator=g, device=dev)
    s = enc.score_all(enc.encode_query(qv), out=torch.empty(n, device=dev))
    s_neg = enc.score_all(enc.encode_query(-qv), out=torch.empty(n, device=dev))
    torch.cuda.synchronize()
    assert torch.equal(s, -s_neg), "complement symmetry: score(q) == -score(~q), exactly"
    zeros = enc.score_all(enc.encode_query(torch.full((dim,), -1.0, device=dev)), out=torch.empty(n, device=dev))
    # q = all-zero bits: xor count = popcount(row); checksum against an independent bit count
    total_bits = 0
    lut8 = torch.tensor([bin(i).count("1") for i in range(256)], device=dev, dtype=torch.int64)
    for lo in range(0, n, 5_000_000):
        total_bits += int(lut8[rows[lo:lo + 5_000_000].long()].sum())
    assert float(((dim - zeros.double()) / 2).sum()) == float(total_bits)
    ids = torch.randint(0, n, (8192,), generator=g, device=dev)
    qbits = enc.encode_query(qv).encoded_vector
    want = qo.bin_score_all(rows[ids].cpu().numpy(), qbits, dim, qo.DOT, False, use_ref=qo.ref() is not None)
    assert_bits_equal(s[ids].cpu().numpy(), want, "sampled rows vs oracle / reference popcount")
    ti, ts = enc.topk(enc.encode_query(qv), 30)
    assert np.array_equal(ts, torch.topk(s, 30).values.cpu().numpy())
    assert np.all(s[torch.from_numpy(ti.astype(np.int64)).to(dev)].cpu().numpy() == ts)
    del enc, rows
    _free()


def test_c4_shard_u8_1536_topk(qo):
    """One of configs[4]'s eight shards: 12.5M x 1536 u8 with top-k (actual_dim > 1040: sums
    can pass 2^24, the default mode is the exact integer rounded once)."""
    n, dim = 12_500_000, 1536
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    data = torch.rand((n, dim), generator=g, device=dev)
    enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, D.Dot, False))
    md = enc.metadata
    q = torch.rand(dim, generator=g, device=dev)
    qobj = enc.encode_query(q)
    s = enc.score_all(qobj, out=torch.empty(n, device=dev))
    ids = torch.randint(0, n, (2048,), generator=g, device=dev)
    rows, meta = qo.u8_encode_with(data[ids].cpu().numpy(), qo.DOT, False, float(md["alpha"]), float(md["offset"]))
    codes, qoff = qo.u8_encode_query(meta, q.cpu().numpy())
    torch.cuda.synchronize()
    assert_bits_equal(s[ids].cpu().numpy(), qo.u8_score_all(meta, rows, codes, qoff, order=qo.ORDER_SIMPLE), "sampled rows")
    ti, ts = enc.topk(qobj, 100)
    order = torch.sort(s, descending=True, stable=True)
    assert np.array_equal(ts, order.values[:100].cpu().numpy())
    assert np.array_equal(ti.astype(np.int64), order.indices[:100].cpu().numpy())
    del order, s
    # configs[4] proper: 1024 queries at once against the shard, top-30 each (the ping-pong MFMA
    # kernel at 1536 dims, 256-query tile), then the 64-query shape (128-query tile).
    _check_batched_topk(qo, enc, data, n, dim, g, dev, n_queries=1024)
    _check_batched_topk(qo, enc, data, n, dim, g, dev, n_queries=64)
    del enc, data
    _free()


def _check_batched_topk(qo, enc, data, n, dim, g, dev, n_queries, k=30):
    """topk_batch at full size: >= 8 sampled queries against the exact single-query topk (ids and
    score bits), 2 of them also against oracle scores recomputed for the winning rows."""
    md = enc.metadata
    queries = torch.rand((n_queries, dim), generator=g, device=dev)
    batch = enc.encode_query_batch(queries)
    ids_d = torch.empty(n_queries * k, dtype=torch.int32, device=dev)
    sc_d = torch.empty(n_queries * k, dtype=torch.float32, device=dev)
    enc.topk_batch(batch, k, out_ids=ids_d, out_scores=sc_d)
    torch.cuda.synchronize()
    ids = ids_d.cpu().numpy().view(np.uint32).reshape(n_queries, k)
    sc = sc_d.cpu().numpy().reshape(n_queries, k)
    picks = sorted({0, n_queries - 1, *np.random.default_rng(n_queries).integers(0, n_queries, 8).tolist()})
    assert len(picks) >= 8 or n_queries < 8
    qobj = None
    for j, qi in enumerate(picks):
        qobj = enc.encode_query(queries[qi], reuse=qobj)
        wi, ws = enc.topk(qobj, k)
        assert np.array_equal(ids[qi], wi), f"query {qi}: ids differ from the single-query top-k"
        assert_bits_equal(sc[qi], ws, f"query {qi}: scores")
        if j < 2:  # the winners' scores straight from the oracle
            win = torch.from_numpy(ids[qi].astype(np.int64)).to(dev)
            rows, meta = qo.u8_encode_with(data[win].cpu().numpy(), qo.DOT, False, float(md["alpha"]), float(md["offset"]))
            codes, qoff = qo.u8_encode_query(meta, queries[qi].cpu().numpy())
            order = qo.ORDER_AVX2 if md["actual_dim"] <= 1040 else qo.ORDER_SIMPLE
            assert_bits_equal(sc[qi], qo.u8_score_all(meta, rows, codes, qoff, order=order), f"query {qi} vs oracle")
    assert np.all(np.diff(sc.astype(np.float64), axis=1) <= 0), "every list is sorted best-first"


def test_c1_batched_topk_10m_x_768(qo):
    """The batched path at the headline store: 10M x 768, 1024 queries (256-query tile) and 64
    queries (128-query tile), top-30 each."""
    n, dim = 10_000_000, 768
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(21)
    data = torch.rand((n, dim), generator=g, device=dev)
    enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, D.Dot, False))
    _check_batched_topk(qo, enc, data, n, dim, g, dev, n_queries=1024)
    _check_batched_topk(qo, enc, data, n, dim, g, dev, n_queries=64)
    del enc, data
    _free()


def test_c4_shard_pq_1536_m192(qo):
    """configs[4]'s PQ leg on one shard: 12.5M x 1536, chunk 8 -> m = 192: the LUT (192 KiB) does
    not fit the LDS, the scan runs in two slices carrying lane sums in the reference's order."""
    n, dim, chunk = 12_500_000, 1536, 8
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(17)
    cen = np.random.default_rng(17).random((256, dim), dtype=np.float32)
    rows = torch.randint(0, 256, (n, 192), generator=g, device=dev, dtype=torch.uint8)
    vp = qa.VectorParameters(dim, n, D.Dot, False)
    enc = qa.EncodedVectorsPQ.from_storage(rows, vp, chunk, cen)
    query = np.random.default_rng(18).random(dim, dtype=np.float32)
    q = enc.encode_query(query)
    lut = qo.pq_encode_query(query, chunk, cen, qo.DOT, False)
    assert_bits_equal(q.lut, lut, "LUT")
    s = enc.score_all(q, out=torch.empty(n, device=dev))
    ids = torch.randint(0, n, (8192,), generator=g, device=dev)
    ids[:2] = torch.tensor([0, n - 1], device=dev)
    torch.cuda.synchronize()
    want = qo.pq_score_all(rows[ids].cpu().numpy(), lut, order=qo.ORDER_SSE)
    assert_bits_equal(s[ids].cpu().numpy(), want, "sampled rows vs oracle (SSE order), sliced LUT")
    ti, ts = enc.topk(q, 30)
    best = torch.sort(s, descending=True, stable=True)
    assert np.array_equal(ts, best.values[:30].cpu().numpy())
    assert np.array_equal(ti.astype(np.int64), best.indices[:30].cpu().numpy())
    del enc, rows, s, best
    _free()


def test_c4_shard_pq_batched_topk_12m5_x_1536_m192(qo):
    """configs[4]'s PQ leg, batched: qamd_pq_topk_batch at 12.5M x 1536 (m = 192, sliced LUT), 24 queries:
    eight sampled queries against the exact single-query top-k (ids and score bits), the winners' scores
    against the oracle's SSE-order LUT sum (encoded_vectors_pq.rs:405-440)."""
    n, dim, chunk, nq, k = 12_500_000, 1536, 8, 24, 30
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(27)
    cen = np.random.default_rng(27).random((256, dim), dtype=np.float32)
    rows = torch.randint(0, 256, (n, 192), generator=g, device=dev, dtype=torch.uint8)
    enc = qa.EncodedVectorsPQ.from_storage(rows, qa.VectorParameters(dim, n, D.Dot, False), chunk, cen)
    queries = np.random.default_rng(28).random((nq, dim), dtype=np.float32)
    ids, sc = enc.topk_batch(enc.encode_query_batch(queries), k)
    assert np.all(np.diff(sc.astype(np.float64), axis=1) <= 0)
    for j, qi in enumerate(sorted({0, nq - 1, *np.random.default_rng(5).integers(0, nq, 8).tolist()})):
        wi, ws = enc.topk(enc.encode_query(queries[qi]), k)
        assert np.array_equal(ids[qi], wi), f"query {qi}: ids differ from the single-query top-k"
        assert_bits_equal(sc[qi], ws, f"query {qi}: scores")
        if j < 3:
            lut = qo.pq_encode_query(queries[qi], chunk, cen, qo.DOT, False)
            win = torch.from_numpy(ids[qi].astype(np.int64)).to(dev)
            assert_bits_equal(sc[qi], qo.pq_score_all(rows[win].cpu().numpy(), lut, order=qo.ORDER_SSE), f"query {qi} vs oracle")
    del enc, rows
    _free()


def test_c3_binary_batched_topk_64q_50m_x_1024(qo):
    """configs[3]'s store with 64 queries at once (bin_gemm_rs_kernel: bits expanded to 0/1 bytes, int8 MFMA, 64-bit
    row offsets on 6.4 GB): sampled queries against the exact single-query top-k and against the oracle's scores of
    the winners computed with the REFERENCE's compiled popcount kernel (impl_xor_popcnt_sse_uint128) when oracle/_ref
    is present; plus, for one query, a full-store check that nothing better than the list's worst entry was missed."""
    n, dim, nq, k = 50_000_000, 1024, 64, 30
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(33)
    rows = torch.randint(0, 256, (n, 128), generator=g, device=dev, dtype=torch.uint8)
    enc = qa.EncodedVectorsBin.from_storage(rows, qa.VectorParameters(dim, n, D.Dot, False))
    queries = torch.randn((nq, dim), generator=g, device=dev)
    batch = enc.encode_query_batch(queries)
    ids, sc = enc.topk_batch(batch, k)
    assert np.all(np.diff(sc.astype(np.float64), axis=1) <= 0)
    qobj = None
    for j, qi in enumerate(sorted({0, 31, 32, nq - 1, *np.random.default_rng(9).integers(0, nq, 6).tolist()})):
        qobj = enc.encode_query(queries[qi], reuse=qobj)
        wi, ws = enc.topk(qobj, k)
        assert np.array_equal(ids[qi], wi), f"query {qi}: ids differ from the single-query top-k"
        assert np.array_equal(sc[qi].view(np.uint32), ws.view(np.uint32)), f"query {qi}: scores"
        if j < 3:
            win = torch.from_numpy(ids[qi].astype(np.int64)).to(dev)
            want = qo.bin_score_all(rows[win].cpu().numpy(), qobj.encoded_vector, dim, qo.DOT, False, use_ref=qo.ref() is not None)
            assert np.array_equal(sc[qi].view(np.uint32), want.view(np.uint32)), f"query {qi} vs oracle / reference popcount"
        if j == 0:  # nothing in the store beats the list's last entry unless it is in the list (ties: lower id wins)
            s_all = enc.score_all(qobj, out=torch.empty(n, device=dev))
            better = torch.nonzero(s_all > float(sc[qi][-1])).flatten().cpu().numpy()
            assert set(better.tolist()) <= set(ids[qi].tolist())
            del s_all
    del enc, rows
    _free()


def test_c0_u8_100k_x_128_full_compare(qo):
    """configs[0]: 100k x 128 f32 -> scalar u8 + dot, every row against the reference CPU SIMD path
    (the oracle's loop over the reference's own compiled impl_score_dot_avx when oracle/_ref is
    present) and every encoded byte against the oracle's encode."""
    n, dim = 100_000, 128
    rng = np.random.default_rng(0)
    data = rng.random((n, dim), dtype=np.float32)
    for dist, dist_id in ((D.Dot, qo.DOT), (D.L2, qo.L2)):
        enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, dist, False))
        rows, meta = qo.u8_encode(data, dist_id, False)
        assert np.array_equal(enc.storage_bytes(), rows), "encoded rows"
        md = enc.metadata
        for key in ("alpha", "offset", "multiplier"):
            assert np.float32(md[key]).view(np.uint32) == np.float32(getattr(meta, key)).view(np.uint32)
        for seed in (1, 2):
            query = np.random.default_rng(seed).random(dim, dtype=np.float32)
            q = enc.encode_query(query)
            codes, qoff = qo.u8_encode_query(meta, query)
            assert np.array_equal(q.encoded_query, codes) and np.float32(q.offset).view(np.uint32) == np.float32(qoff).view(np.uint32)
            want = qo.u8_score_all(meta, rows, codes, qoff, order=qo.ORDER_AVX2, use_ref=qo.ref() is not None)
            assert_bits_equal(enc.score_all(q), want, "all 100k scores vs the reference CPU path")
            hi, _ = qo.topk_heap(-want, 30)  # the caller's heap keeps the SMALLEST of what postprocess() feeds it
            gi, gs = enc.topk(q, 30, largest=True)
            assert np.array_equal(np.sort(gs), np.sort(want[hi])), "top-30 score multiset == the caller's heap"
            strict = want[hi] > gs.min()  # rows strictly better than the boundary score: same ids
            assert set(hi[strict].tolist()) == set(gi[gs > gs.min()].tolist())
