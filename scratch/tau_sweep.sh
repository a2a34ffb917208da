# Lab: the tier threshold of the clustered expansion (csrc/beam.hip, "two tiers") against the step time of
# configs[3] / configs[2] / configs[4]; the rows and levels must not move.  bash scratch/tau_sweep.sh [cfg4 cfg3 cfg5]
cd $GRAFT_REPO_ROOT
out=gpurun_out/tau_sweep
mkdir -p $out
cfgs=${@:-cfg4}
python scratch/cfg_beam.py $cfgs --expansion=single > $out/single.json 2> $out/single.err
for f in 0.01 0.03 0.1 0.3 1 3 10 100; do
  DRT_BEAM_TAU_FACTOR=$f python scratch/cfg_beam.py $cfgs > $out/tau_$f.json 2> $out/tau_$f.err
done
python - <<'P'
import glob, json
for f in sorted(glob.glob("gpurun_out/tau_sweep/*.json")):
    for line in open(f):
        d = json.loads(line)
        print(f.split("/")[-1], d["config"], "step %.4f" % d["s_per_step"], "expand %.1f ms" % d.get("expand_last_ms", -1),
              "levels", d.get("levels"), "rows", d.get("rows"), "paths", d["valid_paths"])
P
