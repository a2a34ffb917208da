# kernel-time breakdown of one beam-pruned step (rocprofv3 --kernel-trace --stats), usage: prof_beam.sh cfg4|cfg3|cfg5 [extra args of cfg_beam.py]
cfg=${1:-cfg4}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_beam_$cfg
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o beam -- python scratch/cfg_beam.py $cfg "$@" > $out/run.log 2>&1
python - "$out" <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
f=glob.glob(out+'/**/beam_kernel_stats.csv', recursive=True)
rows=list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms (2 steps: warm-up + timed)")
for r in rows[:14]:
    print(f"{float(r['TotalDurationNs'])/1e6:10.2f} ms {int(r['Calls']):6d} calls  {float(r['AverageNs'])/1e3:10.1f} us avg  {r['Name'][:110]}")
PY
tail -3 $out/run.log
