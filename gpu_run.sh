# round-5 call 12: which RCCL collectives can be captured in a hipGraph here (one rank)
mkdir -p gpurun_out/r05l
timeout 900 python tools/probes/rccl_graph_probe.py > gpurun_out/r05l/rccl_graph_probe.log 2>&1
cat gpurun_out/r05l/rccl_graph_probe.log
