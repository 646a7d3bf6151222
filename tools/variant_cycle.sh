for r in 1 2; do
for v in old new; do
  if [ $v = old ]; then export RC_HIP_LIB=$PWD/realcamnet_amd/_variants/lib_old.so; else unset RC_HIP_LIB; fi
  echo "$v $(timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*')"
done
done
