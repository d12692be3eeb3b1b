# scratch: the command list of the current experiment (rewritten per gpurun call; nothing depends on it)
echo "nothing queued"
