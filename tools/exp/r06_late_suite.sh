cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-120 > gpurun_out/r06_gputests_late.txt
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 >> gpurun_out/r06_gputests_late.txt
timeout 1500 python tests/tools/fuzz_campaign.py 1500 1010 > gpurun_out/r06_fuzz_campaign_late.txt 2>&1
tail -4 gpurun_out/r06_fuzz_campaign_late.txt >> gpurun_out/r06_gputests_late.txt
cat gpurun_out/r06_gputests_late.txt
