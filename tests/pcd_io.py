"""Minimal readers for the two output files (tests only)."""
import numpy as np


def read_pcd_ascii(path):
    hdr = {}
    with open(path) as f:
        while True:
            line = f.readline()
            if not line:
                raise ValueError("no DATA line")
            if line.startswith("#"):
                continue
            k, _, v = line.strip().partition(" ")
            hdr[k] = v
            if k == "DATA":
                break
        data = np.loadtxt(f, ndmin=2) if int(hdr["POINTS"]) else np.zeros((0, len(hdr["FIELDS"].split())))
    return hdr, data


def read_meta_csv(path):
    with open(path) as f:
        header = f.readline().rstrip("\n")
        rows = np.loadtxt(f, delimiter=",", ndmin=2)
    return header, rows
