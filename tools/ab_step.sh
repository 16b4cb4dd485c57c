for rep in 1 2; do for t in base new; do
  if [ $t = base ]; then export VQK_LIB=/root/repo/scratch/libvqk_base.so; else unset VQK_LIB; fi
  echo -n "$t: "; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['final_loss'])"
done; done
