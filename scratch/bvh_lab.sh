# A/B builds of the LBVH walk (DRT_EXTRA_FLAGS variants separated by |), bench_queries.py per variant
IFS='|' read -ra V <<< "${VARIANTS:-}"
for v in "${V[@]}"; do
  echo "=== variant: [$v]"
  touch differt_amd/csrc/bvh.hip differt_amd/csrc/trace.hip
  DRT_EXTRA_FLAGS="$v" python -m differt_amd.build > /dev/null 2>&1 || echo BUILD FAILED
  python bench_queries.py 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k,v in d.items():
    print(k, {a:('%.3e'%b) for a,b in v.items() if 'bvh' in a or 'sbr_order3_rays' in a})"
done
