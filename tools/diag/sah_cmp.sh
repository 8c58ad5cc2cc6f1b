for b in reference sah; do
  for w in c2 c4; do
    extra=""; if [ $w = c4 ]; then extra="--steps 2 --warmup 6"; fi
    timeout -k 10 300 python bench.py --workload $w --builder $b --no-cpu-baseline --no-c5 $extra > gpurun_out/sah_${w}.json 2> gpurun_out/sah_${w}.err || { tail -3 gpurun_out/sah_${w}.err; exit 1; }
    python3 -c "
import json;d=json.loads(open('gpurun_out/sah_${w}.json').read().strip().splitlines()[-1]);c=d['roofline']['work']['counts'];print('$b','$w',d['ms_per_step'],d['value'],'trace',d['roofline']['kernel_ms'],'nv/ray',round(c['node_visits']/c['rays'],1),'tt/ray',round(c['tri_tests']/c['rays'],2))"
  done
done
