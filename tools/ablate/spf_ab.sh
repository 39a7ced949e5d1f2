for f in tools/ablate/lib_d*.so; do echo "== $f"; RRRMC_HIP_LIB=$PWD/$f timeout 300 python tools/bench_models.py spf 8192 65536 262144 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l); print(d['replicas'], '%.1f ms'%d['kernel_ms'], '%.3e'%d['attempts_per_s'])
    except Exception as e: print(l.strip()[:200])
"; done
