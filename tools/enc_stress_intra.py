import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools')); sys.path.insert(0, ROOT)
import enc_stress_gpu as E
E.CASES[:] = [(1920, 1080, 16, 27, 21, None, 2), (1280, 720, 30, 22, 8, None, 3), (3840, 2160, 4, 32, 5, None, 2), (832, 480, 40, 37, 3, None, 1)]
sys.argv = ['x']
sys.exit(E.main())
