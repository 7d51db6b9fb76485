"""Per-kernel ms/step and calls/step from a rocprofv3 rocpd database (tools/experiments: profiling aid).
usage: python tools/experiments/kstats.py DB STEPS [name-filter]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2])
flt = sys.argv[3] if len(sys.argv) > 3 else ""
rows = list(db.execute("select name, count(*), sum(end-start) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
n = sum(r[1] for r in rows)
print(f"total {tot / 1e6 / steps:.3f} ms/step, {n / steps:.0f} launches/step, {len(rows)} kernels")
for name, cnt, t in rows:
    if flt and flt not in name:
        continue
    short = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", name)
    short = re.sub(r"\(anonymous namespace\)::", "", short)[:80]
    print(f"{t / 1e6 / steps:8.3f} ms/step {cnt / steps:7.1f} calls/step  avg {t / cnt / 1e3:7.1f} us  {short}")
