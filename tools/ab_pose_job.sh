cd "$GRAFT_REPO_ROOT"; O=gpurun_out
EAO_LIB_PATH=gpurun_ab/libeaofusion_hip_head.so python3 tools/ab_pose_bits.py dump $O/pose_head.npz 2>&1 | tail -1
python3 tools/ab_pose_bits.py dump $O/pose_new.npz 2>&1 | tail -1
python3 tools/ab_pose_bits.py cmp $O/pose_head.npz $O/pose_new.npz
timeout -k 10 300 python3 -m pytest tests/test_gpu_lm.py tests/test_gpu_track.py -x -q -k "pose or track or Pose" 2>&1 | tail -3
python3 tools/dbg_pose_waves.py > $O/r05_pose_stamps_after1.txt 2>&1; grep -v "waves=8" $O/r05_pose_stamps_after1.txt | grep -A2 "waves=4"
EAO_LIB_PATH=gpurun_ab/libeaofusion_hip_head.so python3 tools/dbg_pose_waves.py 2>&1 | grep "waves=4"
