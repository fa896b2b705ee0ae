for k in "5-14-4-16" "3-16-6-8" "2-12-9-64"; do
  echo "== $k"; timeout 60 python -m pytest tests/test_gpu_stage2.py -q -m gpu -k "walk_equals and $k" 2>&1 | tail -2
done
echo "== pack"; timeout 60 python -m pytest tests/test_gpu_stage2.py -q -m gpu -k "walk_device_pack" 2>&1 | tail -2
