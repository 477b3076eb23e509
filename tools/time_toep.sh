# GPU box: kernel 2f's three BASELINE shapes (single focus at 256^3, configs[3], configs[1]) in both correction modes: kernel ms, name
for corr in auto fp16; do
python bench.py --foci-per-gpu 1 --no-extras --cpu-seconds 0 --steps 500 --corrections $corr 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('single256', '$corr', round(d['roofline']['kernel_ms_avg'],4), round(d.get('mfma_useful') or 0,3), d['config']['kernel'][:60])"
python bench.py --foci-per-gpu 1 --no-extras --cpu-seconds 0 --steps 100 --elements 32x32 --pitch-mm 1.5 --grid 512 --spacing-mm 0.125 --corrections $corr 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c4', '$corr', round(d['roofline']['kernel_ms_avg'],4), round(d.get('mfma_useful') or 0,3), d['config']['kernel'][:60])"
python bench.py --foci-per-gpu 1 --no-extras --cpu-seconds 0 --steps 1000 --grid 128 --spacing-mm 0.5 --corrections $corr 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c2_128', '$corr', round(d['roofline']['kernel_ms_avg'],4), round(d.get('mfma_useful') or 0,3), d['config']['kernel'][:60])"
done
