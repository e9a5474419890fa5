import os, sys
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
sys.argv=['bench.py','--steps','100','--warmup','20','--no-cpu','--chunk','100']+sys.argv[1:]
import runpy
runpy.run_path(os.path.join(ROOT,'bench.py'), run_name='__main__')
