# cfg3 exhaustive step with the default library and with the filter kernel forced to 6 waves/SIMD
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-scaling 2>/dev/null | python -c "
import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=b['paths']; print('default', p['s_per_step'], p['roofline']['kernel_ms'], p['valid_paths'])"
DIFFERT_AMD_LIB=$PWD/differt_amd/lib/libdiffert_amd_w6.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-scaling 2>/dev/null | python -c "
import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=b['paths']; print('waves6 ', p['s_per_step'], p['roofline']['kernel_ms'], p['valid_paths'])"
