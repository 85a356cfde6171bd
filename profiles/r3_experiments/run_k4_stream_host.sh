for SH in 1 0; do for K in 3 4; do
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $K --master-addr 127.0.0.1 --master-port $((29800 + K + 10*SH)) \
      bench.py --shared-gpu --pinned-input --stream-host $SH --gpus $K --batch 200 --steps 10 --warmup 3 --cpu-budget 0 --loop-batches 0 2>/dev/null | tail -1 | \
      python -c "import json,sys; r=json.loads(sys.stdin.read()); print('stream_host=$SH K=$K device', round(r['value']/1e6,1), 'h2h', round(r['host_to_host']['value']/1e6,1), 'ms', round(r['host_to_host']['ms_per_batch_median'],2))"
done; done
