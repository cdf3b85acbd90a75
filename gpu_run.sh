# round 6, call 2: which victim kernel goes wrong under a second stream, and what do the wrong values look like
mkdir -p gpurun_out/r06b
export TMPDIR=/tmp
SECONDS=0
timeout 600 python tools/two_stream_micro.py --reps 30 > gpurun_out/r06b/micro.log 2>&1; echo "micro rc $? at $SECONDS s"
grep -v "amdgpu.ids" gpurun_out/r06b/micro.log | tail -80 | cut -c1-330
