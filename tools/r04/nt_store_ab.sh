for v in nt product nt product; do echo $v; if [ $v = nt ]; then export ODET_LIB_PATH=$PWD/tools/exp/libodet_ntstore.so; else unset ODET_LIB_PATH; fi
timeout 600 python3 tools/e2e_bench.py --batch 30 --steps 8 --warmup 3 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('  e2e b30', round(d['value'],1), 'backbone+neck ms', round(d['ms_backbone_neck_per_batch'],3))"
timeout 600 python3 tools/e2e_bench.py --batch 1 --graph 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('  e2e b1 graph', round(d['value'],1))"
done
