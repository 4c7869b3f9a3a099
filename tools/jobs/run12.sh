cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for b in 256 32; do echo "== batch $b"; timeout 300 python tools/host_time.py $b 2>/dev/null | grep -E "host issue|host cost|cumulative|forward_raw|training_step|collate|loss_backward|_exchange|_optimizer|next_slot|forward_head|record|wait_event" | head -16; done
