#!/bin/bash
# Round-end check at the BASELINE sizes, HIP in every mode against the oracle directly (tests/full_size_oracle.py).
# usage (GPU box): bash tools/full_size_campaign.sh > gpurun_out/full_size.txt
set -e
echo "# configs[1] (Suzanne room, 1920x1080, 64 passes, 4 bounces)"
timeout -k 10 300 python -u tests/full_size_oracle.py 64
echo; echo "# configs[4] (dragon scene, 64 passes, 8 bounces)"
RT_SCENE=dragon timeout -k 10 400 python -u tests/full_size_oracle.py 64
echo; echo "# configs[3] (991 316 triangles, 3840x2160, 32 of 256 passes)"
RT_SCENE=suzanne5 timeout -k 10 400 python -u tests/full_size_oracle.py 32
echo; echo "# configs[2], keyframes at full image size, 8 passes each"
for f in 12 63 111 187 243; do RT_FRAME=$f timeout -k 10 200 python -u tests/full_size_oracle.py 8 | tail -n 1; done
