//! Replacement for the reference's one-line root module src/zvdb.zig (`pub const HNSW = @import("hnsw.zig").HNSW;`):
//! HNSW(f32) is served by the HIP binding, every other element type (test_hnsw.zig:239-273 uses i32 and f64) keeps the
//! CPU implementation.  UNCOMPILED here (no zig toolchain in this image); see hnsw_hip.zig.
pub fn HNSW(comptime T: type) type {
    return if (T == f32) @import("hnsw_hip.zig").HNSW_f32 else @import("hnsw.zig").HNSW(T);
}
