import os, sys, time
sys.path.insert(0, os.getcwd())
from approximatequeryengine_amd import _native as nat
from approximatequeryengine_amd.engine import Engine, make_query
eng = Engine(0); eng.generate_synthetic(1_000_000)
def t(name, f, reps=300):
    for i in range(5): f(i)
    t0 = time.perf_counter()
    for i in range(reps): f(100 + i)
    print(f"{name}: {(time.perf_counter()-t0)/reps*1e6:.1f} us")
t("plan create+close stride (new pct each)", lambda i: eng.plan(make_query(nat.M_MEMORY_STRIDE, 1.0 + i * 1e-3)).close())
t("plan create+close random 10% new seed", lambda i: eng.plan(make_query(nat.M_RANDOM_POINTER, 10.0, seed=i)).close())
t("reduce random 10% new seed", lambda i: eng.reduce(make_query(nat.M_RANDOM_POINTER, 10.0, seed=i)))
t("reduce random 10% same seed", lambda i: eng.reduce(make_query(nat.M_RANDOM_POINTER, 10.0, seed=7)))
t("reduce stride new pct", lambda i: eng.reduce(make_query(nat.M_MEMORY_STRIDE, 1.0 + i * 1e-3)))
t("plan create+close CLT e=0.01 (new R0 each)", lambda i: eng.plan(make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, max_error_percent=0.01, clt_round0=1024 + i, clt_growth=4)).close())
t("reduce CLT new R0", lambda i: eng.reduce(make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, max_error_percent=0.01, clt_round0=2048 + i, clt_growth=4)))
t("reduce CLT e=1 new R0", lambda i: eng.reduce(make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, max_error_percent=1.0, clt_round0=2048 + i, clt_growth=4)))
