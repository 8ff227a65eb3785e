python tools/probes/prefix_prof.py 30
python tools/probes/prefix_prof.py 8
