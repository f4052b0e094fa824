"""Registry of the SIZE-SELECTED branches of the engine: every place where a size of the caller's data (rows, non-zeros, bytes of x, row
lengths, grid size) decides which code path runs -- and the test(s) that cross it at test size.  The reference has none of this (one fixed kernel per
strategy, no tests: SURVEY.md section 4); a drop-in that adds plans and size rules has to test them.  Round 4's regression (17 planes once x >= 496 MB)
sat behind such a rule that only an out-of-suite R-MAT 26 probe crossed.

tests/test_host_logic.py::test_every_size_threshold_names_a_test checks, without a GPU, that
  * every entry's `pattern` still matches its source file (a renamed / removed rule must be re-registered),
  * every test named here exists in tests/,
  * every named constant (`constexpr ... kName = ...`) of the engine's sources is either registered here or listed in NOT_SIZE_RULES with the
    reason it selects nothing by size -- a NEW constant fails the CPU suite until someone decides which of the two it is.
"""

CSRC = "spmv_acc_amd/csrc/"

# (constant or rule, file, regex that must match the source, tests that cross it, what it selects)
SIZE_RULES = [
    ("kSegMaxPlanes", CSRC + "kernels.hpp", r"constexpr int kSegMaxPlanes = 16;",
     ["test_automatic_slab_count_at_its_maximum", "test_slab_planes_never_exceed_what_the_count_kernels_hold"],
     "planes the run-list count kernels hold: column slabs + the whole-row plane of the two-class form"),
    ("slab_kb (seg_auto_slabs)", CSRC + "dispatch.cpp", r"int seg_auto_slabs\(int n\)",
     ["test_automatic_slab_count_at_its_maximum", "test_automatic_slab_passes_through_the_timed_choice"],
     "automatic slab count = bytes of x / 32 MB, 2 .. 16 (16 from x = 496 MB on)"),
    ("plane clamp", CSRC + "tuner.cpp", r"if \(S_cols > kSegMaxPlanes - \(rest_below > 0 \? 1 : 0\)\)",
     ["test_automatic_slab_count_at_its_maximum"], "the round-4 fix: the whole-row plane takes one of the 16"),
    ("hint_min_x_mb", CSRC + "tuner.cpp", r"tun\(kT_hint_min_x_mb\)",
     ["test_automatic_slab_passes_through_the_timed_choice", "test_whole_row_pass_with_gather_hints"],
     "column census (gather hints, automatic slab passes) only from 96 MB of x on"),
    ("hinted gathers' 32-bit offsets", CSRC + "tuner.cpp", r"static_cast<long long>\(A\.n\) \* 8 >= \(1LL << 32\)",
     ["test_x_beyond_the_hinted_gathers_reach"], "no hints where x is 4 GB or more"),
    ("kMaxGridBlocks / max_grid_blocks", CSRC + "kernels.hpp", r"constexpr int kMaxGridBlocks = 8388593;",
     ["test_grid_stride_paths_at_test_size"], "kernels whose grid grows with m stride over the rows beyond this many workgroups"),
    ("kFlatSmallNnz / flat_small_nnz_k", CSRC + "engine_internal.hpp", r"constexpr int kFlatSmallNnz = 24 << 20;",
     ["test_flat_size_rules_and_hypersparse_tiles"], "flat: tile size and staging order timed per matrix below 24 Mi non-zeros"),
    ("kFlatMaxTileRows", CSRC + "dispatch.cpp", r"constexpr int kFlatMaxTileRows = 16384;",
     ["test_flat_size_rules_and_hypersparse_tiles"], "flat: a tile owning more rows hands the matrix to the row-block kernel"),
    ("kFlatFinish", CSRC + "kernels.hpp", r"constexpr int kFlatFinish = 128;",
     ["test_flat_finish_and_carry_modes"], "flat: a cut row overhanging its tile by at most this is finished by the tile, else carries + fix-up"),
    ("kRowblockMaxRounds", CSRC + "kernels.hpp", r"constexpr int kRowblockMaxRounds = 8;",
     ["test_parity_all_strategies", "test_row_digest_rule_and_its_long_row_escape"], "row blocks: a block needing more LDS rounds sends the matrix to row-block-plus"),
    ("row digest rule (<= 8 per row)", CSRC + "dispatch.cpp", r"static_cast<long long>\(p\.A\.count\(\)\) <= 8LL \* p\.A\.m",
     ["test_row_digest_rule_and_its_long_row_escape"], "1-byte row lengths instead of rowptr where rows average <= 8 non-zeros"),
    ("row digest escape (> 255)", CSRC + "k_rowblock.hip", r"if \(len > 255 \|\| len < 0\) atomicOr",
     ["test_row_digest_rule_and_its_long_row_escape"], "a block with a row longer than 255 reads rowptr after all"),
    ("stream policy rule (<= 8 per row)", CSRC + "tuner.cpp", r"p\.A\.count\(\)\) <= 8LL \* p\.A\.m \? kStreamPolicyNt : kStreamPolicyDefault",
     ["test_deterministic_switch_is_bitwise_stable_across_processes"], "untimed cache policy: non-temporal streams for short rows"),
    ("kTileSpans / long spans to whole waves", CSRC + "tile_stage.hpp", r"const bool posted = w < kWave && span >= 64 && span > 16 \* w;",
     ["test_parity_all_strategies", "test_long_spans_among_short_rows"], "a row span of >= 64 products (and > 16 per lane) is summed by a whole wavefront"),
    ("kSegPiece", CSRC + "kernels.hpp", r"constexpr int kSegPiece = 512;",
     ["test_slab_segments_match_the_oracle"], "slab passes: runs longer than this are cut into pieces merged afterwards"),
    ("kSegShortRow", CSRC + "k_segment.hip", r"constexpr int kSegShortRow = 32;",
     ["test_automatic_slab_count_at_its_maximum", "test_slab_planes_never_exceed_what_the_count_kernels_hold"],
     "run-list count: rows up to this length are counted by one lane, longer ones by a wavefront"),
    ("kPlusLongChunk", CSRC + "kernels.hpp", r"constexpr int kPlusLongChunk = 2 \* kPlusMinNnz;",
     ["test_device_analysis_matches_reference_goldens", "test_out_of_place_is_bitwise_the_in_place_result"],
     "row-block-plus: rows of >= 2 x MIN_NNZ get dedicated blocks (csr_adaptive_plus_analyze.cpp:13-98)"),
    ("kHintSamples", CSRC + "kernels.hpp", r"constexpr int kHintSamples = 8 << 20;",
     ["test_configs3_rmat25_line_enhance_full_size"], "census: all non-zeros below 8 Mi, a strided sample above"),
    ("kHintBins", CSRC + "kernels.hpp", r"constexpr int kHintBins = 4096;",
     ["test_configs3_rmat25_line_enhance_full_size"], "census histogram: counts of 4095 and more share the last bin"),
    ("nnz / m within int32", CSRC + "plan.cpp", r"nnz > INT_MAX - \(1 << 16\)",
     ["test_largest_int32_nnz", "test_error_codes_and_degenerate_shapes"], "problems beyond int32 tile arithmetic are refused (shard them)"),
    ("kMaxPlans", CSRC + "plan.cpp", r"constexpr size_t kMaxPlans = 1024;",
     ["test_plan_cache_is_lru_bounded"], "the least recently used plan is dropped beyond this many"),
    ("kMaxTimed / trial launches by launch duration", CSRC + "engine_internal.hpp", r"if \(first >= 4\.0f\) \{",
     ["test_first_call_is_bounded_and_later_calls_finish_the_timings"], "trial launches per timing: fewer for long kernels"),
    ("32-bit gather offsets (x32_ok)", CSRC + "kernels.hpp", r"inline bool x32_ok\(const CsrDev &A\) \{ return A\.n > 0 && A\.n <= \(1 << 29\); \}",
     ["test_x_beyond_the_hinted_gathers_reach"], "x of 4 GB and more: the general staging form (64-bit gather addresses), no 16-bit column encoding"),
    ("col16: fewer than 64 chunks", CSRC + "tuner.cpp", r"if \(!x32_ok\(A\) \|\| nchunks < 64\)",
     ["test_record_size_follows_the_escape_statistics", "test_row_shards_without_rebasing"], "no encoding below 64 chunks of 256 non-zeros"),
    ("col16: record size by escapes per chunk", CSRC + "tuner.cpp", r"R = h_stats\[1\] <= limit \? 16 : \(h_stats\[2\] <= limit \? 32 : \(h_stats\[3\] <= limit \? 64 : 0\)\);",
     ["test_record_size_follows_the_escape_statistics", "test_encoding_matches_colindex"], "16 / 32 / 64 ints per chunk record: the smallest that at most 1 % of the chunks overflow (12 / 28 / 60 escapes); none: not encoded"),
    ("col16: first chunk of a view", CSRC + "tuner.cpp", r"const int chunk0 = A\.nnz0 / \(kThreads \* kNnzPerThread\) \* \(kThreads \* kNnzPerThread / kCol16Chunk\);",
     ["test_row_shards_without_rebasing"], "the encoding of an un-rebased row sub-range starts at the flat tile that holds its first non-zero"),
    ("kSlabCopyAfterCalls", CSRC + "kernels.hpp", r"constexpr int kSlabCopyAfterCalls = 32;",
     ["test_automatic_slab_major_copy_and_its_value_guard"], "the automatic slab-major copy is built by the call after the 32nd (or inside spmv_acc_prepare)"),
    ("slab-major copy: free-memory rule", CSRC + "dispatch.cpp", r"free_b < 36ull \* static_cast<size_t>\(p\.A\.count\(\)\)",
     ["test_automatic_slab_major_copy_and_its_value_guard", "test_configs3_rmat25_line_enhance_full_size"], "no copy unless 3 x 12 B per non-zero of device memory are free"),
    ("kRowblockTargetRule / kRowblockTargetAlt", CSRC + "engine_internal.hpp", r"constexpr int kRowblockTargetRule = 1800, kRowblockTargetAlt = 1500;",
     ["test_parity_all_strategies", "test_randomised_shapes_all_strategies", "test_deterministic_switch_is_bitwise_stable_across_processes"],
     "products per row block: the two candidates a plan times in turns (the rule, 1800, under `deterministic`); with rows per block = target / (nnz / m) the tile's fill follows the matrix"),
    ("first non-zero of a view (A.nnz0)", CSRC + "tuner.cpp", r"const int tile0 = A\.nnz0 / stride;",
     ["test_chunk_views_are_sized_by_their_own_non_zeros", "test_row_shard_without_rebasing"], "flat: an un-rebased row sub-range starts at its own first tile"),
]

# named constants that select nothing by the size of the caller's data: geometry fixed at build time, enum values, protocol constants
NOT_SIZE_RULES = {
    "kMaxRounds": "ranking rounds a timer can hold", "kMaxCandidates": "candidates a ranking can hold",
    "kGuardSlots": "guard pool: four times kMaxPlans, never exhausted by live plans",
    "kWave": "wavefront width", "kXcds": "XCD count of the remaps (speed only)", "kDigestSlots": "guard_full partial-digest slots in flight",
    "kTuneFields": "fields of a tune-cache line", "kGuardSamples": "rowptr samples of the stale-plan guard",
    "kDigestMaxParts": "guard_full: workgroups of the digest pass (a grid-stride loop covers any m; test_guard_full_notices_an_edit_between_the_samples)",
    "kStreamPolicyNt": "enum", "kStreamPolicyDefault": "enum", "kStreamPolicyIndexDefault": "enum", "kStreamPolicyValueDefault": "enum",
    "kThreads": "workgroup size", "kNnzPerThread": "tile geometry", "kTile": "tile geometry", "kPlusThreads": "analysis geometry (reference instance)",
    "kPlusR": "analysis geometry", "kPlusMinNnz": "analysis geometry (tunable plus_min_nnz, timed)", "kCol16Chunk": "16-bit column encoding geometry",
    "kHintLineShift": "x line = 16 columns", "kPage": "host page size (pin table)", "kChunk": "staging bounce buffer / col16 chunk", "kNcclFloat64": "RCCL enum",
    "kFlatReduceBuilt": "build option", "kLightRowsPerGroup": "LIGHT geometry", "kWaves": "waves per workgroup",
    "kVectorTarget": "vector tile geometry (a tunable until round 5)", "kValueSamples": "value samples of a plan that holds a copy of the values (min(count, this))", "kPlusNpt": "tile geometry", "kPlusTile": "tile geometry", "kPlusMaxRows": "tile geometry", "kSegTile": "tile geometry",
    "kSegCost": "slab passes: cost units per workgroup (balance only)", "kSegMinCost": "slab passes: cost floor of a run (balance only)",
    "kSegEntries": "slab passes: runs per workgroup (tile capacity; test_slab_segments_match_the_oracle fills it)", "kVecTileRows": "vector tile geometry",
}

SCANNED = [CSRC + f for f in ("kernels.hpp", "engine_internal.hpp", "tile_stage.hpp", "device_utils.hpp", "dispatch.cpp", "tuner.cpp", "plan.cpp", "config.cpp",
                              "c_api.cpp", "shard.cpp", "k_flat.hip", "k_rowblock.hip", "k_plus.hip", "k_segment.hip", "k_vector_row.hip", "k_legacy.hip",
                              "k_col16.hip", "k_hint.hip", "k_slab.hip", "k_guard.hip", "k_analyze.hip")]
