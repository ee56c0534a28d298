#!/usr/bin/env python3
"""The reference's own measurement protocol (demos/src/ann_benchmark.rs:104-166, ann_benchmark_data.rs:93-185,202-220) on this
library: for every query of a test set, `encode_query` + the caller's scan with its 30-entry heap - here ONE `topk(30)` call,
host query in, host ids out - timed per query (min / avg / p95 / p99 / max ms, the reference's index rule), and the kNN accuracy
numbers it prints: same_10 / same_20 / same_30 = how many of the TEN true nearest neighbours (exact f32, the data set's
"neighbors" column) are among the first 10 / 20 / 30 results.  The ann-benchmarks HDF5 files are not reachable here (no network):
the data is a seeded Gaussian mixture whose clusters vary along a few directions (embedding-like), in the reference's two flavours - "angular": rows and
queries L2-normalised by `cosine_preprocess` (:223-230) and scored with Dot, results ordered by 1 - score; "euclidean": L2.
Beside every GPU figure the same protocol through the oracle's CPU loop (score_point per row + the caller's heap, one core)
for a few queries, with the ids compared.  For PQ the accuracy is also measured with centroids trained on a RANDOM
10 000-row sample (what the reference draws, encoded_vectors_pq.rs:300-307) instead of this library's evenly strided one.

    python tools/ann_protocol.py [--rows 1000000] [--dims 128,768] [--queries 200] [--cpu-queries 3] [--out FILE.jsonl]
"""
import sys as _sys
if "--help" in _sys.argv[1:] or "-h" in _sys.argv[1:]:  # every tool answers --help without touching the GPU (tests/test_tools.py)
    print(__doc__)
    _sys.exit(0)
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import quantization_amd as qa  # noqa: E402

D = qa.DistanceType


def timings_summary(ms):
    """print_timings (ann_benchmark_data.rs:202-220): ascending sort, avg, first, last, timings[(len * 0.95) as usize], 0.99."""
    t = sorted(ms)
    n = len(t)
    return {"min_ms": t[0], "avg_ms": sum(t) / n, "p95_ms": t[int(np.float32(n) * np.float32(0.95))],
            "p99_ms": t[int(np.float32(n) * np.float32(0.99))], "max_ms": t[-1], "queries": n}


def mixture(rows, dim, queries, seed, dev):
    """Embedding-like synthetic data: sqrt(rows) cluster centres N(0, 1); inside a cluster the points vary along a few
    directions only (16 shared families of 12 directions each, per-cluster scale in [0.3, 1.0]) plus a little isotropic
    noise (0.03) - true neighbours are then decided by a low-dimensional offset, as in learned embeddings, not by
    dim-dimensional noise in which every point of a cluster is equally far from every other.  Queries are fresh points."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    n_c = max(8, int(rows ** 0.5))
    centres = torch.randn((n_c, dim), generator=g, device=dev)
    scale = 0.3 + 0.7 * torch.rand((n_c, 1), generator=g, device=dev)
    n_fam, rank = 16, 12
    families = torch.randn((n_fam, rank, dim), generator=g, device=dev)
    family_of = torch.randint(0, n_fam, (n_c,), generator=g, device=dev)

    def draw(n):
        out = torch.empty((n, dim), device=dev)
        for lo in range(0, n, 1 << 18):  # bounded temporaries
            m = min(1 << 18, n - lo)
            a = torch.randint(0, n_c, (m,), generator=g, device=dev)
            z = torch.randn((m, 1, rank), generator=g, device=dev)
            along = torch.bmm(z, families[family_of[a]]).squeeze(1)
            out[lo:lo + m] = centres[a] + scale[a] * along + 0.03 * torch.randn((m, dim), generator=g, device=dev)
        return out

    return draw(rows), draw(queries)


def cosine_preprocess(x):
    """ann_benchmark_data.rs:223-230: x /= sqrt(sum x^2) unless the squared length is below f32::EPSILON."""
    length = (x * x).sum(dim=1, keepdim=True)
    return torch.where(length < torch.finfo(torch.float32).eps, x, x / length.sqrt())


def exact_neighbours(data, queries, angular, k=10):
    """The data set's `neighbors`: the k true nearest rows per query in f32 (angular: largest dot; euclidean: smallest L2)."""
    out = []
    sq = (data * data).sum(dim=1)
    for q in queries:
        dots = data @ q
        score = dots if angular else 2.0 * dots - sq  # argmax(2 q.v - |v|^2) = argmin |q - v|^2
        out.append(torch.topk(score, k).indices.cpu().numpy())
    return np.stack(out)


def same_counts(ids, truth):
    """same_count (:232-236) of knn[0..10], knn[0..20], knn[0..30] with the ten true neighbours."""
    t = set(int(x) for x in truth)
    return [len(t & set(int(x) for x in ids[:c])) for c in (10, 20, 30)]


def run_gpu(enc, queries_host, truth, largest):
    ms, same = [], np.zeros(3)
    all_ids = []
    for j, q in enumerate(queries_host):
        t0 = time.perf_counter()
        ids, _ = enc.topk(enc.encode_query(q), 30, largest=largest)  # host query in, host ids out: one search
        ms.append((time.perf_counter() - t0) * 1e3)
        same += same_counts(ids, truth[j])
        all_ids.append(np.asarray(ids))
    res = timings_summary(ms)
    res.update({"same_10": same[0] / len(queries_host), "same_20": same[1] / len(queries_host),
                "same_30": same[2] / len(queries_host)})
    return res, all_ids


def run_cpu(kind, enc, queries_host, truth, largest, gpu_ids, extra):
    """The reference's loop on one host core through the oracle: encode_query, score_point for every row
    (score_all = that loop), the caller's 30-entry heap (qo.topk_heap = ann_benchmark_data.rs:151-167)."""
    from oracle import qoracle as qo
    rows = enc.storage_bytes()
    n = rows.shape[0]
    ms, same, agree = [], np.zeros(3), 0
    for j, q in enumerate(queries_host):
        t0 = time.perf_counter()
        if kind == "u8":
            md = enc.metadata
            vp = md["vector_parameters"]
            meta = qo.Meta(md["actual_dim"], float(md["alpha"]), float(md["offset"]), float(md["multiplier"]), vp.dim, n,
                           int(vp.distance_type), int(vp.invert))
            codes, qoff = qo.u8_encode_query(meta, q)
            scores = qo.u8_score_all(meta, rows, codes, qoff, order=qo.ORDER_AVX2, use_ref=qo.ref() is not None)
        elif kind == "pq":
            vp = enc.vector_parameters
            scores = qo.pq_score_all(rows, qo.pq_encode_query(q, extra["chunk"], enc.centroids, int(vp.distance_type), bool(vp.invert)))
        else:
            vp = enc.vector_parameters
            scores = qo.bin_score_all(rows, qo.bin_encode(q[None, :])[0], vp.dim, int(vp.distance_type), bool(vp.invert),
                                      use_ref=qo.ref() is not None)
        post = (np.float32(1.0) - scores) if largest else scores  # the caller's postprocess: |x| 1.0 - x for Dot (:162-166)
        heap_ids, _ = qo.topk_heap(post, 30)
        ms.append((time.perf_counter() - t0) * 1e3)
        same += same_counts(heap_ids, truth[j])
        # the heap and the device top-k pick the same score multiset; inside the boundary tie group the ids may differ
        agree += int(np.array_equal(np.sort(post[np.asarray(heap_ids, dtype=np.int64)]), np.sort(post[gpu_ids[j].astype(np.int64)])))
    res = timings_summary(ms)
    res.update({"same_10": same[0] / len(ms), "same_20": same[1] / len(ms), "same_30": same[2] / len(ms),
                "topk_scores_equal_the_gpus": agree == len(ms), "cores": 1})
    return res


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dims", default="128,768")
    ap.add_argument("--queries", type=int, default=200)
    ap.add_argument("--cpu-queries", type=int, default=3)
    ap.add_argument("--quantizers", default="u8,pq,binary")
    ap.add_argument("--metrics", default="angular,euclidean")
    ap.add_argument("--pq-chunk", type=int, default=8)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    out = open(args.out, "w") if args.out else None

    def emit(rec):
        line = json.dumps(rec)
        print(line, flush=True)
        if out:
            out.write(line + "\n")
            out.flush()

    for dim in [int(x) for x in args.dims.split(",")]:
        for metric in args.metrics.split(","):
            angular = metric == "angular"
            data, queries = mixture(args.rows, dim, args.queries, 2026 + dim, dev)
            if angular:
                data, queries = cosine_preprocess(data), cosine_preprocess(queries)
            dist = D.Dot if angular else D.L2
            largest = angular  # Dot: the caller sorts by 1 - score, i.e. the largest scores first; L2: the smallest
            truth = exact_neighbours(data, queries, angular)
            q_host = queries.cpu().numpy()
            vp = qa.VectorParameters(dim, args.rows, dist, False)
            base = {"protocol": "ann_benchmark (demos/src/ann_benchmark_data.rs:93-185)", "rows": args.rows, "dim": dim,
                    "metric": metric, "distance_type": dist.name, "data": "seeded Gaussian mixture with low-rank clusters"}
            for kind in args.quantizers.split(","):
                variants = []
                t0 = time.perf_counter()
                if kind == "u8":
                    variants.append(("u8", qa.EncodedVectorsU8.encode(data, vp), {}))
                    variants.append(("u8 quantile 0.99", qa.EncodedVectorsU8.encode(data, vp, 0.99), {}))
                elif kind == "pq":
                    extra = {"chunk": args.pq_chunk}
                    enc = qa.EncodedVectorsPQ.encode(data, vp, args.pq_chunk, max_kmeans_threads=8)
                    variants.append((f"pq chunk {args.pq_chunk}, centroids trained on the strided sample", enc, extra))
                    g = torch.Generator(device=dev)
                    g.manual_seed(7)
                    pick = torch.randperm(args.rows, generator=g, device=dev)[: min(10_000, args.rows)].sort().values  # (:300-307)
                    cen = qa.EncodedVectorsPQ.find_centroids(data[pick].contiguous(), args.pq_chunk, 8)
                    variants.append((f"pq chunk {args.pq_chunk}, centroids trained on a RANDOM sample (the reference's draw)",
                                     qa.EncodedVectorsPQ.encode(data, vp, args.pq_chunk, centroids=cen), extra))
                else:
                    variants.append(("binary", qa.EncodedVectorsBin.encode(data, vp), {}))
                torch.cuda.synchronize()
                encode_s = time.perf_counter() - t0
                for name, enc, extra in variants:
                    for _ in range(3):
                        enc.topk(enc.encode_query(q_host[0]), 30, largest=largest)
                    gpu, gpu_ids = run_gpu(enc, q_host, truth, largest)
                    rec = dict(base, quantizer=name, gpu=gpu, encode_seconds_all_variants=round(encode_s, 3))
                    if args.cpu_queries > 0:
                        rec["cpu_oracle_loop"] = run_cpu(kind, enc, q_host[: args.cpu_queries], truth, largest, gpu_ids, extra)
                    emit(rec)
                del variants
            del data, queries
    if out:
        out.close()


if __name__ == "__main__":
    main()
