import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from approximatequeryengine_amd import aqe_backend as m
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
db = m.CustomBPlusDB()
n = 10_000_000
rows = np.zeros(n, dtype=[("id","<i8"),("amount","<f8"),("region","<i4"),("product_id","<i4"),("timestamp","<i8")])
rows["id"] = np.arange(1, n+1); rng = np.random.default_rng(1); rows["amount"] = 1 + 999*rng.random(n); rows["region"] = np.arange(n) % 4; rows["product_id"] = np.arange(n) % 100
db.insert_array(rows)
def t(name, f, reps=2000):
    for _ in range(20): f()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    print(f"{name}: {(time.perf_counter()-t0)/reps*1e6:.1f} us per call")
t("sum_amount (exact)", db.sum_amount, 500)
t("sum_amount_where", lambda: db.sum_amount_where(250.0, 750.0), 500)
t("approx_sum stride 1%", lambda: db.approx_sum(method="stride", sample_percent=1.0))
t("approx_avg clt e=1", lambda: db.approx_avg(method="clt", error_percent=1.0))
t("fast_aggregated_memory_stride_sum 1%", lambda: db.fast_aggregated_memory_stride_sum(1.0))
t("approx_group_by", lambda: db.approx_group_by("avg", "region", 10.0), 500)
s = m.CustomApproximateScheduler()
s.insert_array(rows[:1_000_000])
t("scheduler.execute_sum_query 10%", lambda: s.execute_sum_query("SELECT SUM(amount) FROM sales", 10.0))
