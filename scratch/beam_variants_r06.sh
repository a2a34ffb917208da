# Lab variants of the pruned search as separate libraries (differt_amd/lib/variants/lib_<name>.so), built HERE (hipcc
# cross-compiles): beam.hip with -DDRT_LAB -D<flag>, linked with the objects of the normal build.  Select one on the GPU box
# with DIFFERT_AMD_LIB=differt_amd/lib/variants/lib_<name>.so.
set -e
cd "$(dirname "$0")/.."
mkdir -p differt_amd/lib/variants
OBJ=differt_amd/lib/obj_libdiffert_amd
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fno-slp-vectorize -fno-gpu-rdc -Wno-unused-function"
for v in "$@"; do
  name=$(echo $v | tr -d ' ' | tr 'A-Z' 'a-z' | sed 's/-d//g; s/beam_lab_//g')
  ( /opt/rocm/bin/hipcc $FLAGS -DDRT_LAB $v -x hip -c differt_amd/csrc/beam.hip -o differt_amd/lib/variants/beam_$name.o
    objs=$(ls $OBJ/*.o | grep -v /beam.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o differt_amd/lib/variants/lib_$name.so differt_amd/lib/variants/beam_$name.o $objs -lgomp
    rm differt_amd/lib/variants/beam_$name.o ) &
done
wait
ls -la differt_amd/lib/variants/
