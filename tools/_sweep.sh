for P in 1 2 3 4; do for B in 64 128 256; do
echo "parts $P batch $B: $(ORBHIP_PARTS=$P timeout -k 10 100 python bench.py --no-hamming --no-cpu --batch $B 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')"
done; done
