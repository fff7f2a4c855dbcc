#!/bin/bash
# Known-bytes calibration of FETCH_SIZE / WRITE_SIZE on the FFM kernels' access patterns, plus VALU
# issue costs (tools/fetch_calib.hip).  Separate --pmc passes, as the microarch guide prescribes.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-fetch_calib}
mkdir -p $OUT
[ -x tools/fetch_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/fetch_calib tools/fetch_calib.hip
./tools/fetch_calib 3 > $OUT/plain.jsonl 2> $OUT/plain.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o c -- ./tools/fetch_calib 2 > $OUT/fetch.jsonl 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o c -- ./tools/fetch_calib 2 > $OUT/write.jsonl 2> $OUT/write.err
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcc -o c -- ./tools/fetch_calib 2 > $OUT/tcc.jsonl 2> $OUT/tcc.err
python3 tools/fetch_calib_summary.py $OUT
