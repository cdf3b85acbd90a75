# round-5 call 6: which operation class, run by OTHER processes, makes the norm kernels of process 0 irreproducible
mkdir -p gpurun_out/r05f
for hog in conv1x1_nn conv1x1_wgrad rfft irfft legendre dhconv chan_gemm_f32 chan_wgrad_f32 bias_gelu; do
  RACE_HUNT_TAG=$hog RACE_HUNT_HOG_FILTER=$hog timeout 200 python tools/race_hunt.py --procs 4 --reps 120 --only instnorm > gpurun_out/r05f/hog_$hog.log 2>&1
  echo "== hog $hog rc $?"; grep "RACE\|done\|hog '" gpurun_out/r05f/hog_$hog.log | sort | cut -c1-220
done
