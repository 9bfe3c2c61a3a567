for lib in libpcr.so libpcr_w5.so libpcr_w6.so libpcr_w8.so; do for o in 4.5; do
PCR_LIB=$PWD/pointcloudregistration_amd/$lib timeout -k 10 200 python -u bench.py --steps 20 --warmup 3 --no-cpu-baseline --occupancy $o 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib occ',round(d['index']['mean_occupancy'],2),'it/s',round(d['value'],1),'step_ms',round(d['roofline']['kernel_ms_avg'],3),'knn',{k:round(v/1e9,3) for k,v in d['knn_queries_per_s'].items()})" || exit 1
done; done
