# L2 / L1 request counters of the LBVH first-hit walk (one --pmc pass per set, counters only), usage: pmc_bvh.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_bvh
rm -rf $out && mkdir -p $out
for set in "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $out -o $tag -- python bench_queries.py > $out/$tag.log 2>&1
done
python - "$out" <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
for f in sorted(glob.glob(out+'/**/*counter_collection.csv', recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'bvh' not in k and 'first_hit' not in k: continue
        acc[k[:70]][r['Counter_Name']]+=float(r['Counter_Value']); n[(k[:70],r['Counter_Name'])]+=1
    for k,v in acc.items():
        print(k, {c:(x/ n[(k,c)]) for c,x in v.items()}, 'launches', max(n[(k,c)] for c in v))
PY
