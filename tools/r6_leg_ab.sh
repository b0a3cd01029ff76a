for flags in "--no-config5-leg --no-host-stream-leg --no-channels-leg" "--no-config5-leg --no-host-stream-leg" "--no-host-stream-leg --no-channels-leg" "--no-config5-leg --no-channels-leg"; do
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-per-bins $flags 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['sparse']['streams']['dense_packets']
print('$flags', '->', s['value'], s['ms_per_2^28'], s['steady_state_ms_per_2^28'])"
done
