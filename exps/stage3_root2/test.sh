#!/bin/bash
# Inference on an image folder with the MI355X path (same flags as the reference's exps/stage3_root2/test.sh).
# One GPU:    bash test.sh
# One node:   NGPU=8 bash test.sh      (one process per GPU, frames split in contiguous blocks, RCCL gather of the records)
set -e
export PROJECT_HOME=${PROJECT_HOME:-$(cd "$(dirname "$0")/../.." && pwd)}
export PYTHONPATH=$PYTHONPATH:$PROJECT_HOME
export HSA_ENABLE_IPC_MODE_LEGACY=0
SMAP_MODEL=${SMAP_MODEL:-/path/to/SMAP_model.pth}
REFINE_MODEL=${REFINE_MODEL:-/path/to/RefineNet.pth}
IMAGES=${IMAGES:-/path/to/custom/image_dir}
ARGS=(-p "$SMAP_MODEL" -t run_inference -d test -rp "$REFINE_MODEL" --batch_size 16 --do_flip 1 --dataset_path "$IMAGES")
cd "$(dirname "$0")"
# one process per GPU on a shared host: bound the OpenMP / MKL pools of every rank (the submit thread itself is one core)
NCPU=$(nproc)
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-$(( NCPU / ${NGPU:-1} > 16 ? 16 : (NCPU / ${NGPU:-1} > 0 ? NCPU / ${NGPU:-1} : 1) ))}
if [ "${NGPU:-1}" -gt 1 ]; then
  python -m torch.distributed.run --nnodes=1 --nproc-per-node "$NGPU" --master-addr 127.0.0.1 test.py "${ARGS[@]}"
else
  python test.py "${ARGS[@]}"
fi
