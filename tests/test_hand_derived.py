"""Paper-derived fixtures (tests/golden/hand_derived.json): <= 6 points, m = 2, 2-D integer coordinates, squared-L2 distances
that are small integers.  They are NOT held by the reference's tests -- the reference has no such cases -- so they do not
change what pins the oracle (tests/test_oracle_pins.py does that); they were derived by hand from src/hnsw.zig, line by
line, BEFORE running any implementation, and are run against BOTH the oracle (CPU) and the HIP path (GPU): a misreading of
hnsw.zig shared by the two restatements would have to be shared by the derivation below as well.

Notation: Lx = connections[0] of node x; d(a, b) = squared L2 (hnsw.zig:182-192, no sqrt).

Fixture A (all levels 0; m = 2) -- forces a two-hop greedy descent, a shrinkConnections that drops the entry point from a
list (leaving an asymmetric edge), a frontier heap holding three entries, and a search that returns fewer than k:
  P0 (0,0): first node -> entry point (:110-112).
  P1 (10,0): curr = d(P1,P0) = 100 (:86); L0 empty -> no move; connect(1,0,0) (:106-108): L1=[0], L0=[1].
  P2 (19,0): curr = 361; scan L0=[1]: d(P2,P1) = 81 < 361 -> ep = 1 (:97-100); next while-iteration scans L1=[0]
     (the node current at the TOP of the iteration, :92): 361 < 81 no -> stop.  connect(2,1,0): L2=[1], L1=[0,2].
  P3 (29,0): curr = 841; L0=[1]: 361 -> ep = 1; L1=[0,2]: d(P3,P0) = 841 no, d(P3,P2) = 100 -> ep = 2; L2=[1]: 361 no.
     TWO hops 0 -> 1 -> 2.  connect(3,2,0): L3=[2], L2=[1,3].
  P4 (11,1): curr = 122; L0=[1]: d = 2 -> ep = 1; L1=[0,2]: 122 no, d(P4,P2) = 65 no.  connect(4,1,0): L4=[1];
     L1=[0,2,4] has 3 > m entries -> shrinkConnections(1) (:143-170): distances from P1 = 100, 81, 2 -> stable insertion
     sort ascending -> [4,2,0] -> keep m = 2: L1=[4,2].  L0 still holds 1: the edge 0 -> 1 is now one-way.
  P5 (0,10): curr = 100; L0=[1]: d(P5,P1) = 200 no.  connect(5,0,0): L5=[0], L0=[1,5].
  search((15,0), k=4) (:194-236): seed (0, 225).  pop 0 -> result [0]; push 1 (25), 5 (325).  pop 1 -> result [0,1]; L1=[4,2]:
     push 4 (17), 2 (16): the heap now holds THREE entries {5:325, 4:17, 2:16}.  pop 2 (16) -> push 3 (196).  pop 4 (17) ->
     result has k = 4 entries -> stop (:211).  Final stable sort by distance (:227-233): ids [2,4,1,0], d [16,17,25,225].
     k = 1: the first pop is the entry point itself: [0], [225] (the reference returns the first k POPS, not the k nearest).
     k = 9: everything reachable: pops 0,1,2,4,3,5 -> sorted [2,4,1,3,0,5]; 6 results < k is legal (test_hnsw.zig:122-125).

Fixture B (levels 0,1,1,0,2,1; m = 2) -- forces the level rules of insert / connect:
  P0 (0,0) level 0: entry point; max_level = 0.
  P1 (4,0) level 1: the layer loop runs over the PRE-insert max_level (:88, :114-116): layer 0 only.  connect(1,0,0):
     P1.c[0]=[0], P0.c[0]=[1].  P1.c[1] stays [] (a new tallest node is linked only on layers <= the old max).  max_level = 1.
  P2 (8,0) level 1: curr = 64.  layer 0: P0.c[0]=[1]: d = 16 -> ep = 1; P1.c[0]=[0]: 64 no.  connect(2,1,0): P2.c[0]=[1],
     P1.c[0]=[0,2].  layer 1 (ep, curr carry over): P1.c[1]=[] nothing.  connect(2,1,1): P2.c[1]=[1], P1.c[1]=[2].
  P3 (9,0) level 0: curr = 81.  layer 0: ->1 (25) -> P1.c[0]=[0,2]: d(P3,P2) = 1 -> ep = 2; P2.c[0]=[1]: no.  connect(3,2,0):
     P3.c[0]=[2], P2.c[0]=[1,3].  layer 1: P2.c[1]=[1]: 25 < 1 no; layer 1 > level 0 -> no connect (:106).
  P4 (8,1) level 2: curr = 65; pre-insert max_level = 1 -> layers 0..1.  layer 0: ->1 (17) -> P1.c[0]=[0,2]: d(P4,P2) = 1 ->
     ep = 2; P2.c[0]=[1,3]: 17 no, d(P4,P3) = 2 no.  connect(4,2,0): P4.c[0]=[2]; P2.c[0]=[1,3,4] -> shrink from P2: 16, 1, 1
     -> stable: [3,4,1] (3 before 4: equal distances keep list order) -> [3,4].  layer 1: P2.c[1]=[1]: 17 no.  connect(4,2,1):
     P4.c[1]=[2], P2.c[1]=[1,4].  P4.c[2]=[] .  max_level = 2.
  P5 (9,1) level 1: curr = 82; layers 0..2.  layer 0: ->1 (26) -> ->2 (2) -> P2.c[0]=[3,4]: d(P5,P3) = 1 -> ep = 3; d(P5,P4) = 1
     < 1 no (strict '<': the earliest entry wins a tie, :97); P3.c[0]=[2]: 2 no.  connect(5,3,0): P5.c[0]=[3], P3.c[0]=[2,5].
     layer 1: ep = 3 has no layer 1 (:93) -> no scan; layer 1 <= level 1 -> connect(5,3,1): the source has the layer ->
     P5.c[1]=[3]; the target does NOT (:131 is false) -> nothing appended on its side: an ASYMMETRIC edge to a node that
     lacks the layer.  layer 2: ep = 3 lacks it; 2 > level.
  search((9,2), k=3): seed (0, 85); pop 0 -> push 1 (29); pop 1 -> push 2 (5); pop 2 -> 3 results.  sorted: [2,1,0], [5,29,85].

Fixture C (all levels 0; m = 8: no shrink) -- EXACTLY EQUAL distances in the frontier: the pop order is decided by the internals of
Zig's std.PriorityQueue (std/priority_queue.zig, 0.13: `add` = append + siftUp "while child_index > 0: if compareFn(child, parent) !=
.lt break; move the parent down"; `remove` = removeIndex(0): take items[0], put the LAST item at the root, siftDown "lesser child =
left; the right one only if compareFn(right, left) == .lt; if compareFn(target, lesser child) == .lt break; move the child up").
The reference pins nothing here (its one test with ties, test_hnsw.zig:104-126, has two identical points and asserts only that
both come back); this derivation follows the published std algorithm by hand, heap array written as [id(distance) ...]:
  P0 (0,0) entry.  P1 (2,0): curr 4, L0 empty -> connect(1,0): L1=[0], L0=[1].  P2 (-2,0): curr 4; L0=[1]: d(P2,P1) = 16 no ->
  connect(2,0): L2=[0], L0=[1,2].  P3 (0,2): curr 4; 8, 8 no -> L3=[0], L0=[1,2,3].  P4 (0,-2): curr 4; 8, 8, 16 no -> L4=[0],
  L0=[1,2,3,4].
  search((0,0), k=5): seed 0(0); pop 0 -> result [0]; its list in order: add 1(4): [1]; add 2(4): child 1 / parent 0: 4 < 4 no ->
     [1,2]; add 3(4): parent 0: no -> [1,2,3]; add 4(4): child 3 / parent 1: no -> [1,2,3,4].
     pop -> 1: last = 4 to the root, len 3: children 2(4), 3(4): right < left? no -> lesser = 2; target 4 < 2? (4 < 4) no -> 2 moves up,
        hole at index 1, no children -> [2,4,3].  (1's list = [0]: visited.)
     pop -> 2: last = 3 to the root, len 2: child 4(4): 3 < 4? (4 < 4) no -> 4 moves up -> [4,3].
     pop -> 4, then 3.  Pops 0,1,2,4,3; the final stable sort keeps that order among equal distances: ids [0,1,2,4,3], d [0,4,4,4,4]
     -- node 4 BEFORE node 3 although it entered the heap after it.  k = 4: [0,1,2,4]; k = 3: [0,1,2].
  search((1,0), k=5): seed 0(1); pop 0; add 1(1): [1]; add 2(9): [1,2]; add 3(5): [1,2,3]; add 4(5): child 3 / parent 1 = 2(9): 5 < 9
     -> 2 moves down; parent 0 = 1(1): no -> [1(1),4(5),3(5),2(9)].
     pop -> 1: last 2(9) to the root, len 3: children 4(5), 3(5): right < left no -> lesser 4(5); 9 < 5 no -> 4 up; -> [4(5),2(9),3(5)].
     pop -> 4: last 3(5) to the root, len 2: child 2(9): 5 < 9 -> stop -> [3(5),2(9)].  pop -> 3, pop -> 2.
     Pops 0,1,4,3,2 = sorted already: ids [0,1,4,3,2], d [1,1,5,5,9].
"""
import json
import os

import numpy as np
import pytest

FIX = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hand_derived.json")))


def _check(name, insert, neighbors, search, max_level):
    f = FIX[name]
    for p, lv in zip(f["points"], f["levels"]):
        insert(np.asarray(p, dtype=np.float32), lv)
    assert max_level() == f["max_level"]
    for node, layers in f["adj"].items():
        for layer, want in layers.items():
            got = neighbors(int(node), int(layer))
            assert got is not None and [int(x) for x in got] == want, (name, node, layer, got)
    for s in f["searches"]:
        ids, dist = search(np.asarray(s["query"], dtype=np.float32), s["k"])
        assert [int(x) for x in ids] == s["ids"], (name, s, ids)
        assert [float(x) for x in dist] == [float(x) for x in s["dist"]], (name, s, dist)


@pytest.mark.parametrize("name", ["A", "B", "C"])
def test_oracle_reproduces_the_hand_derived_fixture(oracle, name):
    h = oracle.HNSW(FIX[name]["m"], 200)
    _check(name, lambda p, lv: h.insert(p, lv), h.neighbors, h.search, lambda: h.max_level)
    # a node never has lists above its level (hnsw.zig:19-23)
    for i, lv in enumerate(FIX[name]["levels"]):
        assert h.node_level(i) == lv and h.neighbors(i, lv + 1) is None


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["A", "B", "C"])
@pytest.mark.parametrize("binding", ["device_layers", "steps", "insert_many"])
def test_hip_path_reproduces_the_hand_derived_fixture(name, binding, tmp_path):
    """Through the C ABI: the whole insert on the device (zvdb_hip_insert_batch, one point per call and all points in one
    call) and the step-by-step binding (ZVDB_HOST_INSERT=steps: layer 0 fused, upper layers a host loop) -- in a child
    process, because the choice of binding is read once per process."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import json, sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "import zvdb_amd as zv\n"
        "from tests.test_hand_derived import FIX, _check\n"
        "name, binding = sys.argv[1], sys.argv[2]\n"
        "f = FIX[name]\n"
        "h = zv.HNSW(f['m'], 200)\n"
        "if binding == 'insert_many':\n"
        "    h.insert_many(np.asarray(f['points'], dtype=np.float32), np.asarray(f['levels'], dtype=np.int32))\n"
        "    ins = lambda p, lv: None\n"
        "else:\n"
        "    ins = lambda p, lv: h.insert(p, _level=lv)\n"
        "_check(name, ins, h.connections, h.search_ids, lambda: h.max_level)\n"
        "for i in range(len(f['points'])):\n"
        "    assert [int(x) for x in h.index.get_neighbors(i)] == f['adj'][str(i)]['0']\n"
        "print('fixture ok')\n") % root
    from conftest import tuning_prelude
    pre = tuning_prelude({"ZVDB_HOST_INSERT": "steps"} if binding == "steps" else {})
    r = subprocess.run([sys.executable, "-c", pre + code, name, binding], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "fixture ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
