export TMPDIR=/tmp
mkdir -p gpurun_out/r06s
timeout 300 python tools/tile_stream_probe.py > gpurun_out/r06s/tile_stream.log 2>&1; echo rc $?; grep -v amdgpu.ids gpurun_out/r06s/tile_stream.log | tail -10 | cut -c1-400
