mkdir -p gpurun_out/r5g
P='import json,sys
d=[json.loads(l) for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]
c=d.get("comm")
print("%-22s ms_median %.4f" % (sys.argv[1], d["ms_median"]) + ((" exposed %.4f nocomm %.4f buckets %d" % (c["exposed_ms"], d["ms_per_step_no_comm"], c["buckets"])) if c else ""))'
for i in 1 2; do
python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "$P" plain
python bench.py --force-allreduce --no-cpu-baseline --no-roofline 2>/dev/null | python -c "$P" forced-normal/normal
WUN_SIDE_PRIO=low WUN_COMM_PRIO=high python bench.py --force-allreduce --no-cpu-baseline --no-roofline 2>/dev/null | python -c "$P" forced-low/high
WUN_COMM_PRIO=high python bench.py --force-allreduce --no-cpu-baseline --no-roofline 2>/dev/null | python -c "$P" forced-normal/high
WUN_SIDE_PRIO=low python bench.py --force-allreduce --no-cpu-baseline --no-roofline 2>/dev/null | python -c "$P" forced-low/normal
done 2>&1 | tee gpurun_out/r5g/force_allreduce.txt
