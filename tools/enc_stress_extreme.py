import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools')); sys.path.insert(0, ROOT)
import enc_stress_gpu as E
E.CASES[:] = [(416, 240, 4, 51, 3, None, 100), (416, 240, 3, 4, 5, None, 100), (832, 480, 3, 12, 9, None, 100), (64, 64, 5, 30, 2, None, 100), (128, 64, 4, 26, 4, None, 3), (1920, 1080, 3, 10, 6, None, 100), (64, 192, 3, 30, 2, None, 100), (320, 64, 5, 28, 3, None, 100)]
sys.argv = ['x']
sys.exit(E.main())
