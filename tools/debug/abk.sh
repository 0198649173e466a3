for i in 1 2; do for ck in int real; do for b in 64 256; do for v in main ab1; do
 if [ $v = main ]; then unset BXH_LIB; else export BXH_LIB=$PWD/baryonyx_amd/csrc/libbxh_$v.so; fi
 python bench.py --config c5 --cost-kind $ck --batch $b --steps 40 --warmup 8 --cpu-seconds 0 --e2e-rounds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$ck', $b, '$v', round(d['value']/1e6,2), round(d['roofline']['avg_launch_ms'],4))"
done; done; done; done
