"""Runs a command and prints its wall time and the CPU seconds (user + system) its process tree consumed: python3 tools/cpu_seconds.py <label> <cmd ...>"""
import os, subprocess, sys, time
t0 = time.time()
c0 = os.times()
r = subprocess.run(sys.argv[2:])
c1 = os.times()
print("%s wall %.2f s user %.2f s sys %.2f s" % (sys.argv[1], time.time() - t0, c1.children_user - c0.children_user, c1.children_system - c0.children_system))
sys.exit(r.returncode)
