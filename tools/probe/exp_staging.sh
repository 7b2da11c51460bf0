cd "${GRAFT_REPO_ROOT:-.}"
python3 -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
run() { env "$@" timeout 300 python tools/probe/resident_epoch.py 30 2>&1 | grep "resident epoch"; }
run X=0
run ADVMIL_COPY_PRIORITY=1
run ADVMIL_COPY_PRIORITY=-1
run ADVMIL_STAGE_BLOCKS=128
run ADVMIL_STAGE_BLOCKS=192
run ADVMIL_STAGE_BLOCKS=128 ADVMIL_COPY_PRIORITY=1
run ADVMIL_STAGE_BLOCKS=192 ADVMIL_STAGE_ABLATE=planes ADVMIL_COPY_PRIORITY=1
run ADVMIL_STAGE_ABLATE=skip
run X=0
