cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_sc_$c
  timeout 300 rocprofv3 --pmc $c -d /tmp/pmc_sc_$c -o r --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/pmc_run.py f16r resnet18 > /tmp/pmc_sc_$c.log 2>&1
  f=$(find /tmp/pmc_sc_$c -name '*counter_collection.csv' | head -1)
  python3 - "$f" $c <<'PY'
import csv, sys, collections
rows = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2] and ("shortcut" in r["Kernel_Name"] or "stem" in r["Kernel_Name"]):
        rows[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
for k, v in rows.items():
    mul = 2 * 1024 if sys.argv[2] == "FETCH_SIZE" else 1024
    print(sys.argv[2], k, len(v), [round(x * mul / 1e6, 1) for x in v[:6]], "MB")
PY
done
