# the filter bases beyond Morlet and the general-K DISCO kernels: new GPU tests, then the whole DISCO / FourCastNet3 test files
mkdir -p gpurun_out/r05b
SECONDS=0
timeout 150 python -m pytest tests/test_gpu_disco.py tests/test_fcn3.py -q -m gpu -k "other_bases or piecewise or zernike" > gpurun_out/r05b/new.log 2>&1; echo "new tests rc $? in $SECONDS s"
tail -15 gpurun_out/r05b/new.log
timeout 120 python -m pytest tests/test_gpu_disco.py tests/test_fcn3.py -x -q -m gpu > gpurun_out/r05b/files.log 2>&1; echo "files rc $? at $SECONDS s"
tail -3 gpurun_out/r05b/files.log
