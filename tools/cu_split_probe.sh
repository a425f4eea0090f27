for cfg in "256 256" "224 32" "208 48" "192 64" "176 80"; do set -- $cfg
RT_EXTEND_CUS=$1 RT_SHADE_CUS=$2 timeout -k 10 120 python bench.py --steps 2 --warmup 1 --cpu-spp 0 > gpurun_out/cus_$1_$2.json 2>/dev/null || exit 1
python -c "
import json,sys; d=json.load(open('gpurun_out/cus_$1_$2.json')); r=d['roofline']; print('ext CUs $1 shade CUs $2: extend %.1f ms shade %.1f ms step %.1f' % (r['ms_extend_per_step'], r['ms_shade_per_step'], d['ms_per_step']))"
done
