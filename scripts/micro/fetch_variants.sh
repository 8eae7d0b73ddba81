cd /tmp && export TMPDIR=/tmp
for tag in nosaw base; do
  rm -rf /tmp/pf_$tag
  REPS=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf_$tag -- python $GRAFT_REPO_ROOT/scripts/micro/ffs_variants_bench_proj.py $tag > /dev/null 2>&1
  python3 - $tag <<'PY'
import csv, glob, sys
v=[]
for f in glob.glob(f"/tmp/pf_{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "proj_ffn_split_kernel" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE": v.append(float(r["Counter_Value"]))
print(sys.argv[1], "launches", len(v), "FETCH_SIZE KB avg", sum(v)/max(1,len(v)), "-> MB (x2 gfx950 correction)", 2*sum(v)/max(1,len(v))/1024)
PY
done
