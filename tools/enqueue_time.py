"""Host enqueue time vs GPU time of the fused step (tools only)."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
args = bench.parse_args([]) if hasattr(bench, "parse_args") else None
