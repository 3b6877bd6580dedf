#!/bin/bash
# round 4, tenth GPU call: the same for the Bistro-class frame with measured BRDFs (wavefront kernels; one frame per pass)
bash tools/profile_round.sh r04 m
