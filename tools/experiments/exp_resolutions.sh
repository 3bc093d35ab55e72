for v in 0 2 4; do echo "== kernel_variant=$v"
for sz in 640x360 1280x720 1600x900 1920x1080 2560x1440 3840x2160 800x800 1024x1024 2048x2048 3000x2000; do
  python tools/probe.py --size $sz --variants $v --frames 5 2>&1 | grep -E "atrous|frame wall" | awk -v sz=$sz 'BEGIN{split(sz,a,"x"); mp=a[1]*a[2]/1e6} /frame wall/{fw=$5} /atrous/{t[n++]=$2} END{printf "%-10s %6.2f Mpx  frame %7.1f us  levels:", sz, mp, fw; for(i=0;i<n;i++) printf " %6.1f", t[i]; printf "  (%.1f us/Mpx/level)\n", (t[0]+t[1]+t[2]+t[3]+t[4])/5/mp}'
done; done
