"""Host-side conveniences under the reference's names that are NOT part of the hot path (SURVEY 2.1 rows 16-22:
xyz io, dataset transforms, self-energy estimation, charge normalizers / dipoles, the Assembler).  No kernels, no engine
logic; kept for callers that want the reference's spelling.  Their tests live in tests/extras/."""
