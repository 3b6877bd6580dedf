#!/bin/bash
# round 4, ninth GPU call: rocprofv3 kernel statistics and PMC passes of the final library: Cornell + Sponza-class, then the 10 M triangle scene
bash tools/profile_round.sh r04 1 && bash tools/profile_round.sh r04 2
