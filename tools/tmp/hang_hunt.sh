mkdir -p gpurun_out/hang
for i in 1 2 3 4 5 6; do
  timeout 150 python -X faulthandler -m pytest tests/test_gpu_stage2.py -v -m gpu --timeout 100 --timeout-method=thread > gpurun_out/hang/run$i.log 2>&1
  echo "run $i rc=$? $(tail -1 gpurun_out/hang/run$i.log)"
done
