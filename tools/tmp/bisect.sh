for k in parity_16cam ragged norm_height resident feedback; do
  echo "== $k"; timeout 100 python -m pytest tests/test_gpu_stage2.py -q -m gpu -k "$k or device_pack_matches" 2>&1 | tail -1
done
