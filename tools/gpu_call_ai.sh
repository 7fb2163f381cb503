#!/bin/bash
bash tools/make_profiles_extra.sh r02 2>&1 | tail -3
bash tools/gpu_call_ae.sh
