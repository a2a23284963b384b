for g in 14 22 32 64 90; do RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1920 20 2 exact gen$g | tail -1 || exit 1; RTMI_NO_CULL=1 RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1920 20 2 exact gen$g | tail -1 || exit 1; done
RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1920 8 3 exact sf | tail -1; RTMI_NO_CULL=1 RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1920 8 3 exact sf | tail -1
RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1920 1 3 exact sg | tail -1; RTMI_NO_CULL=1 RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1920 1 3 exact sg | tail -1
