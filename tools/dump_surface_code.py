#!/usr/bin/env python3
"""Lists the surface bytecode of every render call of a GML program (debugging aid).
Usage: tools/dump_surface_code.py scenes/chess.gml"""
import importlib, struct, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
g = importlib.import_module('rust-raytracer_amd.gmlrt')
OPS = "END MOV ADDI SUBI MULI DIVI MODI EQI LESSI REAL FLOOR ADDF SUBF MULF DIVF EQF LESSF SQRT FRAC CLAMPF SIN COS ASIN ACOS SELECT OUT".split()
p = g.Program(path=sys.argv[1])
for ci, c in enumerate(p.calls):
    blob = bytes(c.blob)
    (n_leaves, n_nodes, n_lights, n_code, n_consts, max_stack, max_events, n_bounds) = struct.unpack_from('<8I', blob, 16)
    off_leaves, off_nodes, off_lights, off_code, off_consts, off_bounds = struct.unpack_from('<6Q', blob, 48)
    print(f"call {ci}: leaves {n_leaves} nodes {n_nodes} lights {n_lights} code {n_code} consts {n_consts} stack {max_stack} events {max_events} bounds {n_bounds}")
    code = struct.unpack_from(f'<{n_code}Q', blob, off_code)
    def opnd(x):
        return f"k{x & 0x7fff}" if x & 0x8000 else f"r{x & 63}"
    for i, w in enumerate(code):
        op, dst = w & 0xff, (w >> 8) & 0xff
        a, b, cc = (w >> 16) & 0xffff, (w >> 32) & 0xffff, (w >> 48) & 0xffff
        print(f"  {i:3d}: {OPS[op]:7s} " + ("" if op == 0 else f"{'out' if op == 25 else 'r'}{dst} <- {opnd(a)} {opnd(b)} {opnd(cc)}"))
