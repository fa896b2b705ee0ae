cd $GRAFT_REPO_ROOT
for cfg in "--seqs 128 --contexts 3" "--seqs 256 --contexts 3" "--seqs 512 --contexts 2" "--seqs 512 --contexts 3" "--seqs 1024 --contexts 2"; do
  echo "== $cfg"
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline $cfg 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"
done
