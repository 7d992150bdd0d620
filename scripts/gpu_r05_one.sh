cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=/root/repo/gpurun_out/r05_outlier; mkdir -p $O; rm -f $O/*.txt
F="LV_DEMO_FORCE_REBUILD=80,LV_DEMO_FORCE_REBUILD2=160,LV_SLOW_CALL_MS=1"
LV_STREAM_AB="fa=$F,LV_DEMO_CYCLE_DUMP=$O/fa.txt" timeout 900 python scripts/stream_bench_cpp.py 2>$O/err.txt | tail -1 > $O/stream.json
python - <<PY
rows = [l.split() for l in open("$O/fa.txt")]
for r in rows[30:]:
    if float(r[1]) > 0.9: print("cycle", r)
prev = None
for r in rows:
    if prev and (prev[2] != r[2] or prev[3] != r[3]): print("transition at", r[0], prev[2], "->", r[2], "adopted", r[3])
    prev = r
PY
grep -n "slow call" $O/err.txt | tail -30
