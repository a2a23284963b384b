#!/bin/bash
# Evidence of a round under gpurun_out/<name>/ (copied into profiles/<name>_final/ afterwards):  bash tools/evidence_round.sh <name>
set -o pipefail
R=$PWD
N=${1:-r2c}
O=$R/gpurun_out/$N
bash tools/profile_round.sh $N > gpurun_out/${N}_profile.log 2>&1 || { echo "profile_round failed"; tail -5 gpurun_out/${N}_profile.log; exit 1; }
cd $R
for b in exact strict fast; do
RTMI_LIB=$R/tools/librtmi_stamps.so timeout -k 10 120 python tools/quick.py 1920 125 2 $b > $O/stamps_$b.log 2>&1 || exit 1
done
{
echo "# The five benchmark configurations on one MI355X, kernel time of one launch (tools/quick.py, RT_CHUNK_AUTO), one GPU session."
echo "# 1  3-sphere scene 400x225, 100 spp, depth 50"
for b in exact strict fast; do RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 400 100 3 $b s3 | tail -1 || exit 1; done
echo "# 2  3-sphere scene 1920x1080, 500 spp, depth 50"
for b in exact strict fast; do RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1920 500 3 $b s3 | tail -1 || exit 1; done
echo "# 3  final scene 1200x675, 500 spp"
for b in exact strict fast; do RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1200 500 3 $b sf | tail -1 || exit 1; done
echo "# smaller launches of the final scene, 1920x1080 at 1000 / 125 / 8 spp"
for spp in 1000 125 8; do RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1920 $spp 3 exact sf | tail -1 || exit 1; done
} > $O/configs.txt 2>&1 || { cat $O/configs.txt; exit 1; }
{
echo "# tools/quick.py 1920 20 2 exact genN (rt_scene_generate with half extent N), exact build, RT_CHUNK_AUTO, one GPU session"
for g in 14 16 22 32 45 64 90; do RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1920 20 2 exact gen$g | tail -1 || exit 1; done
} > $O/scene_size_sweep.txt 2>&1 || { cat $O/scene_size_sweep.txt; exit 1; }
timeout -k 10 300 python tools/part_scaling.py > $O/part_scaling.txt 2>&1 || { cat $O/part_scaling.txt; exit 1; }
timeout -k 10 300 python tools/part_scaling2.py >> $O/part_scaling.txt 2>&1 || exit 1
timeout -k 10 200 python tools/progressive_fps.py > $O/progressive_fps.txt 2>&1 || exit 1
RTMI_LIB=$R/tools/librtmi_stamps.so timeout -k 10 300 python tools/part_stamps.py > $O/part_stamps.log 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_parts -o p -- python3 $R/tools/part_scaling.py > $O/pmc_parts.log 2> $O/pmc_parts.err || exit 1
cp $(find $O/pmc_parts -name '*counter_collection.csv' | head -1) $O/pmc_parts.csv && rm -rf $O/pmc_parts
cd $R
grep -v amdgpu $O/configs.txt $O/scene_size_sweep.txt $O/part_scaling.txt $O/progressive_fps.txt | cut -c1-220
