#!/usr/bin/env python3
"""Hierarchy-build throughput of the host mirror (frames per second out of N concurrent builder threads), with a build cut into pool
tasks or on one thread each.  No GPU needed.  Usage: python tools/build_throughput_probe.py"""
import sys, time, numpy as np, concurrent.futures as cf
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from simple_raytracer_amd import host
import golden_util as gu
T = host.Transformation
bunny = gu.load_mesh("bunny")
def build(f):
    om = host.ObjectManager(); om.add_object("bunny", bunny)
    om.transformTriangles("bunny", T.scaleObj(1500.0, 1500.0, 1500.0)); om.transformTriangles("bunny", T.rotateObjX(T.radians(180.0 + f))); om.transformTriangles("bunny", T.changeObjPosition(20.0, 170.0, 300.0))
    om.createBoundingHierarchy("bunny"); return om
for tasks in (True, False):
    host.set_build_tasks(tasks)
    for nb in (1, 4, 8, 12, 16):
        with cf.ThreadPoolExecutor(nb) as ex:
            t0=time.perf_counter(); list(ex.map(build, range(24))); dt=(time.perf_counter()-t0)/24*1e3
        print("tasks", tasks, "builders", nb, "ms/frame", round(dt,2))
