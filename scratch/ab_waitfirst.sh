#!/bin/bash
for lib in libmer.so libmer_waitfirst.so; do
  MER_LIB=$PWD/mitsubaer_amd/$lib ./scratch/ab_quick.sh --options lds_bricks=0
done
