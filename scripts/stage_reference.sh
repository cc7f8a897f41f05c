#!/bin/bash
# Stage the files the reference's driver needs (its gccNMF/*.py and the dev1 mixture) into git-ignored scratch that travels to the GPU
# box with gpurun (oracle/_ref/ is in .gitignore, not in .gpurunignore), run the drop-in test there once, keep the log, remove the copy.
#   bash scripts/stage_reference.sh stage | unstage
set -e
REF=${GCCNMF_REFERENCE_ROOT:-/root/reference}
DST=oracle/_ref/reference_checkout
if [ "$1" = "stage" ]; then
  mkdir -p $DST/gccNMF $DST/data
  cp $REF/gccNMF/*.py $DST/gccNMF/
  cp $REF/data/dev1_female3_liverec_130ms_1m_mix.wav $DST/data/
  echo staged; ls $DST/gccNMF | wc -l
else
  rm -rf $DST; echo unstaged
fi
