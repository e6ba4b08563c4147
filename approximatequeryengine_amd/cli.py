"""Command line front end with the flag semantics the reference documents
(/root/reference/enhanced_aqe_cli.py:388-553, README):

    python -m approximatequeryengine_amd.cli "SELECT SUM(amount) FROM sales" --db sales.db --s 10
    python -m approximatequeryengine_amd.cli "SELECT AVG(amount) FROM sales" --db sales.db --e 2 --ci
    python -m approximatequeryengine_amd.cli "SELECT APPROX(SUM(amount)) FROM sales" --db sales.db --compare
    python -m approximatequeryengine_amd.cli --explain

The reference's own CLI defines `-s/--sample` and `-e/--error` but tests `args.s` / `args.e`
(enhanced_aqe_cli.py:107-110, 412-415), so `--s 10` silently runs the exact query and `--e 2` is rejected as
ambiguous.  Here `--s` and `--e` are real options.  The aggregate and its interval are computed
on the GPU in one call (no list of Python Record objects, enhanced_aqe_cli.py:189-200).
"""
from __future__ import annotations

import argparse
import os
import re
import sys
import time
from typing import Optional, Tuple

QUERY_EXACT, QUERY_RANDOM, QUERY_CLT, QUERY_EMBEDDED = "exact", "random_sample", "clt_approximation", "embedded_approx"

METHODS = {  # enhanced_aqe_cli.py:36-81
    "random": "Strided/random sampling of the given percentage",
    "clt": "Central-limit-theorem monitor with an error threshold",
    "block": "Contiguous blocks of rows",
    "adaptive": "Picks a method from the error requirement",
    "parallel": "Region-per-worker strided sampling",
    "revolutionary": "Method chosen from the table size",
}


def parse_embedded_approx(query: str) -> Tuple[str, bool]:
    """enhanced_aqe_cli.py:83-95: APPROX(func) -> func."""
    pat = r"APPROX\s*\(\s*([^)]+\))\s*\)"
    m = re.search(pat, query, re.IGNORECASE)
    if m:
        return re.sub(pat, m.group(1), query, flags=re.IGNORECASE), True
    return query, False


def aggregate_of(query: str) -> str:
    up = query.upper()
    for a in ("SUM", "AVG", "COUNT"):
        if a + "(" in up:
            return a
    return "AVG"  # enhanced_aqe_cli.py:198-200: default to average


def determine_query_type(query: str, args) -> str:
    """enhanced_aqe_cli.py:97-114 with the attribute names fixed."""
    if parse_embedded_approx(query)[1]:
        return QUERY_EMBEDDED
    if args.s is not None:
        return QUERY_RANDOM
    if args.e is not None:
        return QUERY_CLT
    return QUERY_EXACT


def get_optimal_method_for_query(query: str, dataset_size: Optional[int] = None) -> str:
    """enhanced_aqe_cli.py:116-131."""
    up = query.upper()
    if "SUM(" in up or "COUNT(" in up:
        return "revolutionary" if dataset_size and dataset_size > 100_000 else "clt"
    if "AVG(" in up:
        return "random"
    if "GROUP BY" in up:
        return "parallel"
    return "adaptive"


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(prog="aqe", description="Approximate SUM/AVG/COUNT on MI355X",
                                allow_abbrev=False)
    p.add_argument("query", nargs="?", help="SQL query, e.g. \"SELECT SUM(amount) FROM sales\"")
    p.add_argument("--db", default="custom_demo.db", help="database file (reference format)")
    p.add_argument("-s", "--s", "--sample", dest="s", type=float, metavar="PERCENT", help="sample percentage")
    p.add_argument("-e", "--e", "--error", dest="e", type=float, metavar="THRESHOLD", help="CLT error threshold, percent")
    p.add_argument("--method", choices=list(METHODS), help="override the method")
    p.add_argument("--compare", action="store_true", help="also run the exact query")
    p.add_argument("--explain", action="store_true", help="list the methods")
    p.add_argument("--threads", type=int, default=4, help="pointers/regions (default 4)")
    p.add_argument("--confidence", type=float, default=0.95)
    p.add_argument("--ci", action="store_true", help="print the confidence interval")
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--device", type=int, default=0)
    p.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="under torchrun (one process per GPU): the torch.distributed backend (nccl = RCCL)")
    p.add_argument("--collective", choices=["torch", "mailbox"], default="torch", help="under torchrun: the all-reduce of the moment vectors "
                   "(mailbox = the peer-mapped one-launch form, for GPUs with peer access)")
    return p


def run(args, out=sys.stdout) -> int:
    if args.explain:
        for k, v in METHODS.items():
            print(f"{k:<14}{v}", file=out)
        return 0
    if not args.query:
        print("error: a query is required unless --explain is given", file=out)
        return 2
    if not os.path.exists(args.db):
        print(f"error: database file '{args.db}' not found", file=out)
        return 1
    from . import aqe_backend
    clean, _ = parse_embedded_approx(args.query)
    qtype = determine_query_type(args.query, args)
    agg = aggregate_of(clean)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "torch.distributed" in sys.modules and sys.modules["torch.distributed"].is_initialized():  # called from a program that has its group
        world = sys.modules["torch.distributed"].get_world_size()
    if world > 1:
        # launched one process per GPU (torchrun --nproc-per-node N -m approximatequeryengine_amd.cli ...): the table is sharded
        # by row region over the ranks, every rank runs the same calls, rank 0 reports
        import torch.distributed as dist
        from .sharded_backend import ShardedBPlusDB
        own_group = not dist.is_initialized()
        if own_group:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29544")
            if args.backend == "nccl":
                import torch
                dist.init_process_group("nccl", device_id=torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))))
            else:
                dist.init_process_group(args.backend)
        if dist.get_rank() != 0:
            out = open(os.devnull, "w")
        try:
            db = ShardedBPlusDB(device_id=args.device if args.backend != "nccl" else None, collective=args.collective)
            return _run_on(db, args, out, clean, qtype, agg, aqe_backend, f"{world} GPUs, sharded by row region")
        finally:
            if own_group:
                dist.barrier()
                dist.destroy_process_group()
    db = aqe_backend.CustomBPlusDB(device_id=args.device)
    return _run_on(db, args, out, clean, qtype, agg, aqe_backend, None)


def _run_on(db, args, out, clean, qtype, agg, aqe_backend, sharded_note) -> int:
    if not db.open_database(args.db):
        print(f"error: cannot open database: {args.db}", file=out)
        return 1
    db._path = ""  # a read-only session must not rewrite the file on close
    n = db.get_total_records()
    print(f"query: {args.query}\ndatabase: {args.db} ({n:,} records{', ' + sharded_note if sharded_note else ''})\ntype: {qtype}", file=out)
    t0 = time.perf_counter()
    gb = re.search(r"GROUP\s+BY\s+(region|product_id)\b", clean, flags=re.IGNORECASE)
    if gb:  # one sweep, one (n, S, Q) bin per key, an interval per group (executor.cpp:202-321 semantics)
        pct = args.s if args.s is not None else (100.0 if qtype == QUERY_EXACT else 10.0)
        groups = db.approx_group_by(agg, group_by=gb.group(1), sample_percent=pct, method="exact" if pct >= 100.0 else "rowid",
                                    where=aqe_backend.parse_where(clean))
        ms = (time.perf_counter() - t0) * 1e3
        print(f"\nGROUP BY {gb.group(1).lower()} ({'exact' if pct >= 100.0 else f'rowid sample {pct:g}%'}):", file=out)
        for key, g in groups.items():
            ci = f"   ({g.ci_lower:,.4f} - {g.ci_upper:,.4f})" if (args.ci and pct < 100.0) else ""
            print(f"   {key:>6}: {g.value:,.4f}{ci}   n={g.n:,}", file=out)
        print(f"   execution time: {ms:.2f} ms", file=out)
        db.close_database()
        return 0
    if qtype == QUERY_EMBEDDED:
        method = args.method or get_optimal_method_for_query(clean, n)
        qtype = QUERY_CLT if method == "clt" else QUERY_RANDOM  # enhanced_aqe_cli.py:489-494
        e, s = 2.0, 10.0
    else:
        e, s = args.e if args.e is not None else 5.0, args.s if args.s is not None else 10.0
    # `WHERE amount BETWEEN a AND b` / `>= a AND <= b` / `> a` (the façade's extraction, custom_scheduler.cpp:277-294) is
    # honoured by every sampler that has a WHERE form; the reference CLI ignores it for scalar queries altogether
    where = aqe_backend.parse_where(clean)
    if qtype == QUERY_RANDOM:
        # the reference CLI's own routing by table size when no method is named (enhanced_aqe_cli.py:178-186): memory stride
        # above 50 k rows, direct access above 10 k, the sequential sampler below
        auto = "stride" if n > 50_000 else "direct_access" if n > 10_000 else "sequential"
        m = {"block": "block", "parallel": "region", "random": "random"}.get(args.method or "", auto)
        res = db.approx(agg, method=m, sample_percent=s, seed=args.seed, num_threads=args.threads, where=where)
        name = f"{m} sampling ({s}%)"
    elif qtype == QUERY_CLT:
        if where is not None:
            print("note: the CLT sampler has no WHERE form (clt_validated_dual_pointer_sample samples the whole table): the WHERE clause is ignored, as in the reference CLI", file=out)
            where = None
        res = db.approx(agg, method="clt", error_percent=e, num_threads=args.threads, confidence_level=args.confidence)
        name = f"CLT (±{e}%)"
    else:
        res = db.approx(agg, method="exact", where=where)
        name = "exact"
    ms = (time.perf_counter() - t0) * 1e3
    print(f"\n{name} result:\n   value: {res.value:,.4f}", file=out)
    if (args.ci or qtype == QUERY_CLT) and qtype != QUERY_EXACT:
        print(f"   confidence interval: ({res.ci_lower:,.4f} - {res.ci_upper:,.4f})", file=out)
    print(f"   samples used: {res.n:,}   rounds: {res.rounds}   converged: {bool(res.converged)}", file=out)
    print(f"   execution time: {ms:.2f} ms (kernels {res.kernel_ms * 1e3:.1f} us, {res.achieved_GBps:.0f} GB/s algorithmic)", file=out)
    if args.compare and qtype != QUERY_EXACT:
        exact = db.approx(agg, method="exact", where=where)
        print(f"\ncomparison:\n   approximate: {res.value:,.4f}\n   exact:       {exact.value:,.4f}", file=out)
        if exact.value != 0:
            print(f"   actual error: {abs(res.value - exact.value) / abs(exact.value) * 100:.4f}%", file=out)
    db.close_database()
    return 0


def main(argv=None) -> int:
    return run(build_parser().parse_args(argv))


if __name__ == "__main__":
    raise SystemExit(main())
