#!/usr/bin/env python3
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name'].replace('void (anonymous namespace)::', '').split('(')[0]
    if float(r['Percentage']) > 0.5:
        print('   %-44s calls %4s avg %9.1f us  min %9.1f us' % (n[:44], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
