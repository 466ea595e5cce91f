#!/bin/bash
# round 6, session d: the whole GPU suite on the split libraries (product + exp), smoke, the one-rank RCCL tests
TAG=${TAG:-r06d}
bash tools/gpu_round.sh $TAG smoke tests
