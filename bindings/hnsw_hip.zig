//! hnsw_hip.zig -- drop-in for `HNSW(f32)` of allisoneer/zvdb's src/hnsw.zig over libzvdb_hip (include/zvdb_hip.h).
//!
//! UNCOMPILED: there is no zig toolchain in the image this repository is built in (zig: not found; the reference asks
//! for zig >= 0.13, build.zig.zon:9).  The file is complete -- no elided bodies -- and written against Zig 0.13's std
//! (managed `ArrayList.init(allocator)`, as hnsw.zig:22 uses it); a maintainer compiles it with the reference's own
//! `zig build test` after applying bindings/build.zig.patch.  Its C++ twin, zvdb_amd/host/hnsw.hpp, has the same structure,
//! is compiled and is what tests/cpp/test_hnsw_host.cpp (the reference's src/test_hnsw.zig restated) runs against.
//!
//! Beside the four reference signatures the struct has the two calls through which a device is worth having -- `insertMany`
//! and `searchMany` (ids + distances, no Node copies): one launch per batch instead of one launch + one synchronisation per
//! point / query.  They are the 1:1 twins of `HNSW::insert_many` / `HNSW::search_many` in hnsw.hpp, which are compiled and
//! tested (tests/cpp/test_hnsw_host.cpp: bulk_calls_equal_per_call).
//!
//! What stays in Zig (north_star: "host-side graph bookkeeping stays in Zig"): ids, the owned copy of every point (search
//! results alias it, test_hnsw.zig:67), every layer's adjacency lists, the level RNG, the mutex, error mapping.
//! What runs on the GPU: `distance` (hnsw.zig:182-192) at every call site, the greedy neighbour-list scans
//! (:85-104), `connect` + `shrinkConnections` (:119-170) on the device mirror of every layer, the best-first expansion
//! and result ordering of `search` (:194-236).  Two insert bindings, the choice fixed per index at the first insert:
//!   * device layers (default): ONE call per insert, `zvdb_hip_insert_batch` -- the whole of :85-108 on the device; the
//!     mirror adopts, per layer, the lists the device reports;
//!   * step by step (`use_device_layers = false`): layer 0 fused (`zvdb_hip_insert_layer0`), upper layers a host loop over
//!     `zvdb_hip_distances_from`, `connect` / `shrinkConnections` below -- the call-site-by-call-site binding.
const std = @import("std");
const Allocator = std.mem.Allocator;
const ArrayList = std.ArrayList;
const c = @cImport(@cInclude("zvdb_hip.h"));

pub const HNSW_f32 = struct {
    const Self = @This();

    /// hnsw.zig:12-42.  `connections[l]` is the node's list at layer l (level + 1 lists).
    pub const Node = struct {
        id: usize,
        point: []f32,
        connections: []ArrayList(usize),
        mutex: std.Thread.Mutex,

        fn init(allocator: Allocator, id: usize, point: []const f32, level: usize) !Node {
            const connections = try allocator.alloc(ArrayList(usize), level + 1); // :19
            errdefer allocator.free(connections);
            for (connections) |*conn| conn.* = ArrayList(usize).init(allocator); // :20-23
            const owned = try allocator.alloc(f32, point.len); // :24
            errdefer allocator.free(owned);
            @memcpy(owned, point); // :25
            return Node{ .id = id, .point = owned, .connections = connections, .mutex = std.Thread.Mutex{} };
        }

        fn deinit(self: *Node, allocator: Allocator) void { // :36-41
            for (self.connections) |*conn| conn.deinit();
            allocator.free(self.connections);
            allocator.free(self.point);
        }
    };

    /// `nodes` keeps the reference's field name; ids are dense insertion order (hnsw.zig:77), so it is a list and
    /// `nodes.count()` (test_hnsw.zig:198) is provided below.
    pub const NodeList = struct {
        items: ArrayList(Node),
        pub fn count(self: *const NodeList) usize {
            return self.items.items.len;
        }
        pub fn get(self: *const NodeList, id: usize) ?Node {
            return if (id < self.items.items.len) self.items.items[id] else null;
        }
    };

    allocator: Allocator,
    nodes: NodeList,
    entry_point: ?usize,
    max_level: usize,
    m: usize,
    ef_construction: usize,
    mutex: std.Thread.Mutex,
    dev: ?*c.zvdb_hip_index,
    use_device_layers: bool,

    pub fn init(allocator: Allocator, m: usize, ef_construction: usize) Self { // :52-62 (infallible, no allocation)
        return .{
            .allocator = allocator,
            .nodes = .{ .items = ArrayList(Node).init(allocator) },
            .entry_point = null,
            .max_level = 0,
            .m = m,
            .ef_construction = ef_construction,
            .mutex = std.Thread.Mutex{},
            .dev = null,
            .use_device_layers = true,
        };
    }

    pub fn deinit(self: *Self) void { // :64-71
        for (self.nodes.items.items) |*n| n.deinit(self.allocator);
        self.nodes.items.deinit();
        if (self.dev) |d| c.zvdb_hip_destroy(d);
        self.dev = null;
    }

    fn check(rc: c_int) !void {
        switch (rc) {
            c.ZVDB_OK => {},
            c.ZVDB_ERR_OOM => return error.OutOfMemory,
            c.ZVDB_ERR_NODE_NOT_FOUND => return error.NodeNotFound, // :120-121
            c.ZVDB_ERR_DIM_MISMATCH => @panic("Mismatched dimensions in distance calculation"), // :183-185
            else => return error.DeviceError, // zvdb_hip_last_error() has the text
        }
    }

    fn ensureDevice(self: *Self, dim: usize) !void {
        if (self.dev != null) return;
        var cfg = std.mem.zeroes(c.zvdb_hip_config);
        cfg.abi_version = c.ZVDB_HIP_ABI_VERSION;
        cfg.dim = @intCast(dim); // no dimension is declared at init (:52): the first insert implies it
        cfg.metric = c.ZVDB_METRIC_L2SQ;
        cfg.m = @intCast(self.m);
        cfg.ef_construction = @intCast(self.ef_construction);
        cfg.device = -1;
        var handle: ?*c.zvdb_hip_index = null;
        try check(c.zvdb_hip_create(&cfg, &handle));
        self.dev = handle;
    }

    fn randomLevel() usize { // :172-180, unchanged (unobservable through search: SURVEY 0.4)
        var level: usize = 0;
        const max_level_cap = 31;
        while (level < max_level_cap and std.crypto.random.float(f32) < 0.5) level += 1;
        return level;
    }

    pub fn insert(self: *Self, point: []const f32) !void { // :73-117
        self.mutex.lock(); // :74-75
        defer self.mutex.unlock();
        try self.ensureDevice(point.len);
        const id = self.nodes.count(); // :77
        const level = randomLevel(); // :78
        if (self.use_device_layers) return self.insertDeviceLayers(point, id, level);
        return self.insertStepByStep(point, id, level);
    }

    /// n inserts with the result of n `insert` calls in this order (extension: hnsw.zig:73 takes one point per call).
    /// `points` is n rows of `dim` floats, row-major.  ONE `zvdb_hip_insert_batch` call carries the whole batch -- every layer's
    /// scan, connect and shrinkConnections (:85-108) for all n points inside the device -- and the mirror then adopts, per point
    /// and layer, the lists the device reports (the same records `insertDeviceLayers` reads for one point).  This is where the
    /// device wins: ~470k inserts/s against ~43k/s through per-call `insert` (one launch + one synchronisation each) and
    /// ~200k/s for the reference on one CPU core (100k x 128, INTEGRATION.md 3).  Twin: `HNSW::insert_many` (hnsw.hpp).
    /// Only for an index on the device-layers binding (the default); a step-by-step index takes `insert` per point.
    pub fn insertMany(self: *Self, points: []const f32, dim: usize) !void {
        self.mutex.lock(); // :74-75
        defer self.mutex.unlock();
        if (dim == 0 or points.len % dim != 0) return error.DeviceError;
        const n = points.len / dim;
        if (n == 0) return;
        try self.ensureDevice(dim);
        if (!self.use_device_layers) return error.DeviceError;
        const levels = try self.allocator.alloc(u32, n);
        defer self.allocator.free(levels);
        var total: usize = 0;
        for (levels) |*lv| { // :78, drawn here in insertion order: the binding owns the RNG
            const level = randomLevel();
            lv.* = @intCast(level);
            total += level + 1;
        }
        const rw = 4 + self.m; // record = {ep, owner's list length, ep's list length, ep has the layer, ep's list[m]}
        const nrec = try self.allocator.alloc(u32, n);
        defer self.allocator.free(nrec);
        const recs = try self.allocator.alloc(u32, total * rw);
        defer self.allocator.free(recs);
        var first: u64 = 0;
        try check(c.zvdb_hip_insert_batch(self.dev, points.ptr, @intCast(dim), n, levels.ptr, &first, nrec.ptr, recs.ptr, total));
        std.debug.assert(first == self.nodes.count());
        try self.nodes.items.ensureUnusedCapacity(n);
        var r: usize = 0; // cursor into the records: point i owns nrec[i] of them, lowest layer first
        for (0..n) |i| {
            const id = self.nodes.count(); // :77
            const level: usize = levels[i];
            const node = try Node.init(self.allocator, id, points[i * dim ..][0..dim], level); // :79
            self.nodes.items.appendAssumeCapacity(node); // :82 (the list owns it from here: `deinit` frees it)
            if (self.entry_point != null) { // :84
                for (0..nrec[i]) |l| { // layers 0 .. min(level, max_level): where connect ran (:106)
                    const rec = recs[r * rw ..][0..rw];
                    r += 1;
                    const ep: usize = rec[0];
                    const own = &self.nodes.items.items[id].connections[l];
                    own.clearRetainingCapacity();
                    if (rec[1] != 0) try own.append(ep);
                    if (rec[3] != 0) {
                        const lst = &self.nodes.items.items[ep].connections[l];
                        lst.clearRetainingCapacity();
                        for (rec[4 .. 4 + rec[2]]) |nb| try lst.append(nb);
                    }
                }
            } else {
                self.entry_point = id; // :110-112
            }
            if (level > self.max_level) self.max_level = level; // :114-116
        }
    }

    /// nq searches in ONE device launch (extension: hnsw.zig:194 takes one query per call): for query q the ids and squared-L2
    /// distances `search(queries[q], k)` would return, in its order, in `out_ids[q * k ..][0 .. out_count[q]]` /
    /// `out_dist[...]` (unused slots: maxInt(u64) / +inf; fewer than k results are legal, test_hnsw.zig:122-125).  No `Node`
    /// copies: a caller that wants the points reads `nodes.get(id).?.point`.  Per call through `search` the device answers one
    /// query in ~34 us (a launch + a synchronisation; the reference on one CPU core: 12.5-13.7 us); through this call
    /// it answers ~9M queries per second (100k x 128, k = 10: INTEGRATION.md 3).  Twin: `HNSW::search_many` (hnsw.hpp).
    pub fn searchMany(self: *Self, queries: []const f32, dim: usize, k: usize, out_ids: []u64, out_dist: []f32, out_count: []u32) !void {
        self.mutex.lock(); // :195-196
        defer self.mutex.unlock();
        if (dim == 0 or queries.len % dim != 0) return error.DeviceError;
        const nq = queries.len / dim;
        if (out_ids.len < nq * k or out_dist.len < nq * k or out_count.len < nq) return error.DeviceError;
        @memset(out_count[0..nq], 0);
        @memset(out_ids[0 .. nq * k], std.math.maxInt(u64));
        @memset(out_dist[0 .. nq * k], std.math.inf(f32));
        const entry = self.entry_point orelse return; // :201: an empty index answers nothing
        if (nq == 0 or k == 0) return;
        // ef = 0: the reference-compat expansion (:201-233) per query, one wave each, all queries in one launch
        try check(c.zvdb_hip_search_graph(self.dev, queries.ptr, @intCast(nq), @intCast(dim), @intCast(k), 0, entry, out_ids.ptr, out_dist.ptr, out_count.ptr, null));
    }

    /// The whole of :85-108 in one device call; the mirror adopts what the device settled on.
    fn insertDeviceLayers(self: *Self, point: []const f32, id: usize, level: usize) !void {
        const rw = 4 + self.m; // record = {ep, owner's list length, ep's list length, ep has the layer, ep's list[m]}
        const recs = try self.allocator.alloc(u32, (level + 1) * rw);
        defer self.allocator.free(recs);
        var lv: u32 = @intCast(level);
        var nrec: u32 = 0;
        var first: u64 = 0;
        try check(c.zvdb_hip_insert_batch(self.dev, point.ptr, @intCast(point.len), 1, &lv, &first, &nrec, recs.ptr, level + 1));
        std.debug.assert(first == id);
        var node = try Node.init(self.allocator, id, point, level); // :79
        errdefer node.deinit(self.allocator);
        try self.nodes.items.append(node); // :82
        if (self.entry_point != null) { // :84
            var l: usize = 0;
            while (l < nrec) : (l += 1) { // layers 0 .. min(level, max_level): where connect ran (:106)
                const r = recs[l * rw ..][0..rw];
                const ep: usize = r[0];
                const own = &self.nodes.items.items[id].connections[l];
                own.clearRetainingCapacity();
                if (r[1] != 0) try own.append(ep); // the new node's list is [ep] (an m of 0 shrinks it away)
                if (r[3] != 0) { // ep has this layer (:128-133): its list after the append + shrinkConnections
                    const lst = &self.nodes.items.items[ep].connections[l];
                    lst.clearRetainingCapacity();
                    for (r[4 .. 4 + r[2]]) |nb| try lst.append(nb);
                }
            }
        } else {
            self.entry_point = id; // :110-112
        }
        if (level > self.max_level) self.max_level = level; // :114-116
    }

    /// Call site by call site: layer 0 fused on the device, upper layers on the host lists with batched distances.
    fn insertStepByStep(self: *Self, point: []const f32, id: usize, level: usize) !void {
        var dev_id: u64 = 0;
        var ep: u64 = if (self.entry_point) |e| e else std.math.maxInt(u64);
        var curr_dist: f32 = 0;
        var ep_n: u32 = 0;
        var owner_n: u32 = 0;
        const ep_list = try self.allocator.alloc(u64, self.m + 1);
        defer self.allocator.free(ep_list);
        // Node.init's copy, :86, the layer-0 scan :90-104 and connect(id, ep, 0) with both shrinks, in one launch
        try check(c.zvdb_hip_insert_layer0(self.dev, point.ptr, @intCast(point.len), ep, &dev_id, &ep, &curr_dist, ep_list.ptr, &ep_n, &owner_n, null));
        std.debug.assert(dev_id == id);
        var node = try Node.init(self.allocator, id, point, level);
        errdefer node.deinit(self.allocator);
        try self.nodes.items.append(node);
        if (self.entry_point != null) {
            var layer: usize = 0;
            while (layer <= self.max_level) : (layer += 1) { // :88, the pre-insert max_level
                if (layer == 0) {
                    const own = &self.nodes.items.items[id].connections[0];
                    if (owner_n != 0) try own.append(@intCast(ep));
                    const lst = &self.nodes.items.items[@intCast(ep)].connections[0];
                    lst.clearRetainingCapacity();
                    for (ep_list[0..ep_n]) |nb| try lst.append(@intCast(nb));
                } else {
                    var changed = true;
                    while (changed) { // :90
                        changed = false;
                        const cur = self.nodes.items.items[@intCast(ep)]; // :92 (the node current at the top of the iteration)
                        if (layer < cur.connections.len and cur.connections[layer].items.len > 0) { // :93
                            const lst = cur.connections[layer].items;
                            const ids = try self.allocator.alloc(u64, lst.len);
                            defer self.allocator.free(ids);
                            const dist = try self.allocator.alloc(f32, lst.len);
                            defer self.allocator.free(dist);
                            for (lst, 0..) |nb, i| ids[i] = nb;
                            try check(c.zvdb_hip_distances_from(self.dev, id, ids.ptr, @intCast(ids.len), dist.ptr)); // :96, the whole list (:94)
                            for (ids, dist) |nb, d| {
                                if (d < curr_dist) { // :97 strict: earliest wins ties
                                    ep = nb;
                                    curr_dist = d;
                                    changed = true;
                                }
                            }
                        }
                    }
                }
                if (layer >= 1 and layer <= level) try self.connect(id, @intCast(ep), layer); // :106-108 (layer 0: done on the device)
            }
        } else {
            self.entry_point = id;
        }
        if (level > self.max_level) self.max_level = level;
    }

    /// hnsw.zig:119-141 for layers >= 1 of the step-by-step binding (layer 0 lives in zvdb_hip_insert_layer0).
    fn connect(self: *Self, source: usize, target: usize, level: usize) !void {
        if (source >= self.nodes.count() or target >= self.nodes.count()) return error.NodeNotFound; // :120-121
        const source_node = &self.nodes.items.items[source];
        const target_node = &self.nodes.items.items[target];
        source_node.mutex.lock(); // :123-126 (redundant under the global mutex, kept as in the reference)
        defer source_node.mutex.unlock();
        if (source != target) target_node.mutex.lock();
        defer if (source != target) target_node.mutex.unlock();
        if (level < source_node.connections.len) try source_node.connections[level].append(target); // :128-130
        if (level < target_node.connections.len) try target_node.connections[level].append(source); // :131-133
        if (level < source_node.connections.len) try self.shrinkConnections(source, level); // :135-137
        if (level < target_node.connections.len) try self.shrinkConnections(target, level); // :138-140
    }

    /// hnsw.zig:143-170.  The comparator's two distance() calls per comparison (:160-161) become ONE batched
    /// zvdb_hip_distances_from(owner, candidates); the stable insertion sort then runs on (distance, position) pairs with
    /// the reference's strict `<` (std.sort.insertion is stable, so equal distances keep list order, as :166 does).
    fn shrinkConnections(self: *Self, node_id: usize, level: usize) !void {
        const connections = &self.nodes.items.items[node_id].connections[level];
        if (connections.items.len <= self.m) return; // :146
        const n = connections.items.len;
        const ids = try self.allocator.alloc(u64, n);
        defer self.allocator.free(ids);
        const dist = try self.allocator.alloc(f32, n);
        defer self.allocator.free(dist);
        for (connections.items, 0..) |nb, i| ids[i] = nb;
        try check(c.zvdb_hip_distances_from(self.dev, node_id, ids.ptr, @intCast(n), dist.ptr));
        const Entry = struct { id: usize, dist: f32 };
        const entries = try self.allocator.alloc(Entry, n); // :148-156 (the reference copies the ids; so do we, with distances)
        defer self.allocator.free(entries);
        for (ids, dist, 0..) |nb, d, i| entries[i] = .{ .id = @intCast(nb), .dist = d };
        const Ctx = struct {
            pub fn lessThan(_: void, a: Entry, b: Entry) bool {
                return a.dist < b.dist; // :162
            }
        };
        std.sort.insertion(Entry, entries, {}, Ctx.lessThan); // :166
        connections.shrinkRetainingCapacity(self.m); // :168
        for (entries[0..self.m], 0..) |e, i| connections.items[i] = e.id; // :169
    }

    pub fn search(self: *Self, query: []const f32, k: usize) ![]const Node { // :194-236
        self.mutex.lock(); // :195-196
        defer self.mutex.unlock();
        var result = try ArrayList(Node).initCapacity(self.allocator, k); // :198
        errdefer result.deinit();
        if (self.entry_point) |entry| {
            if (k > 0) {
                const ids = try self.allocator.alloc(u64, k);
                defer self.allocator.free(ids);
                const dist = try self.allocator.alloc(f32, k);
                defer self.allocator.free(dist);
                var count: u32 = 0;
                // frontier heap, visited set, expansion and the final stable sort (:201-233) on the device; ef = 0 selects
                // the reference-compat mode; fewer than k results are legal (test_hnsw.zig:122-125)
                try check(c.zvdb_hip_search_graph(self.dev, query.ptr, 1, @intCast(query.len), @intCast(k), 0, entry, ids.ptr, dist.ptr, &count, null));
                for (ids[0..count]) |i| try result.append(self.nodes.items.items[@intCast(i)]); // by-value Node copies (:213-214): point / connections alias index memory
            }
        }
        return result.toOwnedSlice(); // caller frees with the same allocator (:235)
    }
};
