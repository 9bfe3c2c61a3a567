for t in "0=0,1=1" "0=0,1=2"; do for o in 1.0 2.0 3.0; do
timeout -k 10 200 python -u bench.py --steps 20 --warmup 3 --no-cpu-baseline --occupancy $o --tune "$t" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tune $t occ',round(d['index']['mean_occupancy'],2),'it/s',round(d['value'],1),'step_ms',round(d['roofline']['kernel_ms_avg'],3),'knn',{k:round(v/1e9,3) for k,v in d['knn_queries_per_s'].items()})" || exit 1
done; done
