#!/bin/bash
bash scripts/gpu_r04_g.sh
bash scripts/gpu_r04_h.sh
