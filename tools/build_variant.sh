# A/B builds of one translation unit: `bash tools/build_variant.sh <tag> <file.hip> [extra hipcc flags...]` compiles <file.hip> with the
# extra flags and links it with the current objects of every other unit of the DEFAULT library into robir_amd/librobir_hip_<tag>.so
# (tools/ab_dvis.py loads it).  Per-file flags of the Makefile (-mllvm -amdgpu-mfma-vgpr-form, unroll thresholds) are NOT inherited: pass them.
set -e
cd "$(dirname "$0")/../robir_amd/csrc"
TAG=$1; F=$2; shift 2
make -s -j8 default >/dev/null
B=${F%.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize "$@" -c $F -o build/${B}_${TAG}.ovar
OBJS=$(make -s --no-print-directory -f Makefile -f - print-objs <<'MK'
print-objs:
	@echo $(OBJS)
MK
)
OBJS=$(echo $OBJS | tr ' ' '\n' | grep -v "build/${B}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../librobir_hip_${TAG}.so $OBJS build/${B}_${TAG}.ovar
echo built ../librobir_hip_${TAG}.so
