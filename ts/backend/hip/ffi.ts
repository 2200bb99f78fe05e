// ffi.ts — bun:ffi binding of libtstwo_hip.so (C ABI: include/tstwo_hip.h).
//
// Intended location in tstwo: packages/core/src/backend/hip/ffi.ts.  NOT TESTED in the build image (no Bun there);
// the same ABI, call for call, is exercised by tstwo_amd/_lib.py (ctypes) in every `-m gpu` test of this repository.
// Device pointers travel as u64 (bigint); `P` arguments are host typed arrays (ptr(...)).
import { dlopen, FFIType, ptr, toArrayBuffer } from "bun:ffi";

const { i32, u32, u64, ptr: P, cstring } = FFIType;

const lib = dlopen(process.env.TSTWO_HIP_LIB ?? "libtstwo_hip.so", {
  tstwo_init: { args: [i32], returns: i32 },
  tstwo_shutdown: { args: [], returns: i32 },
  tstwo_last_error: { args: [], returns: cstring },
  tstwo_version: { args: [], returns: cstring },
  tstwo_device_count: { args: [P], returns: i32 },
  tstwo_device_name: { args: [P, u64], returns: i32 },
  tstwo_set_stream: { args: [u64], returns: i32 },
  tstwo_sync: { args: [], returns: i32 },
  tstwo_trim: { args: [], returns: i32 },
  tstwo_set_alloc_mode: { args: [i32], returns: i32 },
  tstwo_graph_begin_capture: { args: [], returns: i32 },
  tstwo_graph_end_capture: { args: [P], returns: i32 },
  tstwo_graph_launch: { args: [u64], returns: i32 },
  tstwo_graph_destroy: { args: [u64], returns: i32 },
  tstwo_event_create: { args: [P], returns: i32 },
  tstwo_event_record: { args: [u64], returns: i32 },
  tstwo_event_elapsed_ms: { args: [u64, u64, P], returns: i32 },
  tstwo_event_destroy: { args: [u64], returns: i32 },
  tstwo_malloc: { args: [P, u64], returns: i32 },
  tstwo_free: { args: [u64], returns: i32 },
  tstwo_upload: { args: [u64, P, u64], returns: i32 },
  tstwo_host_register: { args: [P, u64], returns: i32 },
  tstwo_host_unregister: { args: [P], returns: i32 },
  tstwo_host_alloc: { args: [P, u64], returns: i32 },
  tstwo_host_free: { args: [P], returns: i32 },
  tstwo_upload_async: { args: [u64, P, u64], returns: i32 },
  tstwo_upload_fence: { args: [], returns: i32 },
  tstwo_upload_wait: { args: [], returns: i32 },
  tstwo_download: { args: [P, u64, u64], returns: i32 },
  tstwo_download_many: { args: [P, P, u64, P], returns: i32 },
  tstwo_copy: { args: [u64, u64, u64], returns: i32 },
  tstwo_zero: { args: [u64, u64], returns: i32 },
  tstwo_comm_unique_id: { args: [P], returns: i32 },
  tstwo_comm_init: { args: [i32, i32, P], returns: i32 },
  tstwo_comm_destroy: { args: [], returns: i32 },
  tstwo_comm_info: { args: [P, P], returns: i32 },
  tstwo_allgather_roots: { args: [u64, u64], returns: i32 },
  tstwo_allgather: { args: [u64, u64, u64], returns: i32 },
  tstwo_allgather_async: { args: [u64, u64, u64], returns: i32 },
  tstwo_comm_wait: { args: [], returns: i32 },
  tstwo_m31_add: { args: [u64, u64, u64, u64], returns: i32 },
  tstwo_m31_sub: { args: [u64, u64, u64, u64], returns: i32 },
  tstwo_m31_mul: { args: [u64, u64, u64, u64], returns: i32 },
  tstwo_m31_neg: { args: [u64, u64, u64], returns: i32 },
  tstwo_m31_batch_inverse: { args: [u64, u64, u64], returns: i32 },
  tstwo_cm31_batch_inverse: { args: [P, P, u64], returns: i32 },
  tstwo_qm31_batch_inverse: { args: [P, P, u64], returns: i32 },
  tstwo_m31_batch_inverse_async: { args: [u64, u64, u64], returns: i32 },
  tstwo_cm31_batch_inverse_async: { args: [P, P, u64], returns: i32 },
  tstwo_qm31_batch_inverse_async: { args: [P, P, u64], returns: i32 },
  tstwo_check_zero_flag: { args: [], returns: i32 },
  tstwo_qm31_mul: { args: [P, P, P, u64], returns: i32 },
  tstwo_secure_accumulate: { args: [P, P, u64], returns: i32 },
  tstwo_bit_reverse: { args: [P, u64, u64], returns: i32 },
  tstwo_twiddles_build: { args: [u32, u32, u64, u64], returns: i32 },
  tstwo_cfft_evaluate: { args: [P, u64, u32, u32, u64, u32], returns: i32 },
  tstwo_cfft_interpolate: { args: [P, u64, u32, u32, u64, u32], returns: i32 },
  tstwo_cfft_interpolate_to: { args: [P, P, u64, u32, u32, u64, u32], returns: i32 },
  tstwo_cfft_evaluate_extended: { args: [P, u32, P, u64, u32, u32, u64, u32], returns: i32 },
  tstwo_cfft_plan_passes: { args: [u32, u64, P], returns: i32 },
  tstwo_poly_extend: { args: [u64, u32, u64, u32], returns: i32 },
  tstwo_eval_at_point: { args: [u64, u32, P, P, P], returns: i32 },
  tstwo_eval_at_point_batch: { args: [P, u64, u32, P, P, P], returns: i32 },
  tstwo_line_interpolate: { args: [P, u32, u64, u32, P], returns: i32 },
  tstwo_fri_fold_line: { args: [P, u32, u64, u32, P, P], returns: i32 },
  tstwo_fri_fold_line_tw: { args: [P, u32, u64, P, P], returns: i32 },
  tstwo_fri_fold_circle_into_line: { args: [P, u64, P, u32, u64, u32, P], returns: i32 },
  tstwo_fri_fold_circle_into_line_tw: { args: [P, u64, P, u32, u64, P], returns: i32 },
  tstwo_fri_fold_line_dev: { args: [P, u32, u64, u32, u64, P], returns: i32 },
  tstwo_fri_fold_circle_into_line_dev: { args: [P, u64, P, u32, u64, u32, u64], returns: i32 },
  tstwo_channel_mix_root_draw_felt: { args: [u64, u64, u64], returns: i32 },
  tstwo_fri_fold_line_rows: { args: [P, u32, u64, u64, u64, u32, P, P], returns: i32 },
  tstwo_fri_fold_circle_into_line_rows: { args: [P, P, u32, u64, u64, u64, u32, P], returns: i32 },
  tstwo_fri_decompose: { args: [P, u64, P, P], returns: i32 },
  tstwo_merkle_commit_layer: { args: [u32, u64, P, u64, u64], returns: i32 },
  tstwo_merkle_commit: { args: [P, P, u64, u64, P], returns: i32 },
  // reqs: a packed array of tstwo_commit_request structs (4 x 8 bytes each: cols, log_sizes, n_cols, layers)
  tstwo_merkle_commit_many: { args: [P, u64, P], returns: i32 },
  tstwo_merkle_layers_bytes: { args: [u32], returns: u64 },
  tstwo_merkle_decommit: { args: [u64, u32, P, P, u64, P, P, P, u64, P, P, P, P, P, P], returns: i32 },
  // reqs: a packed array of tstwo_decommit_request structs (9 x 8 bytes each: layers, max_log (u32, padded), cols, col_log_sizes,
  // n_cols, query_logs, queries, n_queries, n_query_sets) written into a BigUint64Array
  tstwo_merkle_decommit_many: { args: [P, u64, P, P, P, P, P], returns: i32 },
  // layers: a packed array of tstwo_fri_layer structs (5 x 8 bytes each: layers, max_log (u32, padded), cols, eval_logs, n_evals)
  // out: a packed array of tstwo_fri_layer_out structs (6 x 8 bytes each: log_size (u32, padded), cols[4], layers); first_tree: one u64
  tstwo_fri_commit_layers: { args: [P, P, u64, u64, u32, u32, u64, u64, u64, P, P, u64, P], returns: i32 },
  tstwo_fri_decommit: { args: [P, u64, P, u64, u32, u32, u32, P, P, P, P, P, P], returns: i32 },
  tstwo_gather_words: { args: [P, P, u32, u64, P], returns: i32 },
  tstwo_grind_blake2s: { args: [P, u32, u64, P], returns: i32 },
  tstwo_quotients_accumulate_samples: { args: [u32, u32, P, u64, u64, P, P, P, P, P, P], returns: i32 },
  tstwo_quotients_accumulate: { args: [u32, u32, P, u64, u64, P, P, P, P, P, P, P, P, P], returns: i32 },
  tstwo_quotients_accumulate_samples_async: { args: [u32, u32, P, u64, u64, P, P, P, P, P, P], returns: i32 },
  tstwo_quotients_accumulate_async: { args: [u32, u32, P, u64, u64, P, P, P, P, P, P, P, P, P], returns: i32 },
});

export const hip = lib.symbols;

/** Throws with the reference's own error text ("0 has no inverse", "length is not power of two", ...). */
export function check(rc: number): void {
  if (rc !== 0) throw new Error(String(hip.tstwo_last_error()));
}

let initialised = false;
export function ensureInit(): void {
  if (!initialised) {
    check(hip.tstwo_init(Number(process.env.LOCAL_RANK ?? 0)));   // one process per GPU
    initialised = true;
  }
}

/** An owned device allocation. */
export class DeviceBuffer {
  readonly dev: bigint;
  constructor(readonly nbytes: number) {
    ensureInit();
    const out = new BigUint64Array(1);
    check(hip.tstwo_malloc(ptr(out), BigInt(Math.max(nbytes, 16))));
    this.dev = out[0]!;
  }
  /** Take ownership of a block the LIBRARY allocated with tstwo_malloc and handed out (tstwo_fri_commit_layers). */
  static adopt(dev: bigint, nbytes: number): DeviceBuffer {
    const b = Object.create(DeviceBuffer.prototype) as { dev: bigint; nbytes: number };
    b.dev = dev; b.nbytes = nbytes;
    return b as DeviceBuffer;
  }
  upload(words: Uint32Array | Uint8Array, byteOffset = 0): void {
    if (words.byteLength) check(hip.tstwo_upload(this.dev + BigInt(byteOffset), ptr(words), BigInt(words.byteLength)));
  }
  downloadU32(count: number, byteOffset = 0): Uint32Array {
    const out = new Uint32Array(count);
    if (count) check(hip.tstwo_download(ptr(out), this.dev + BigInt(byteOffset), BigInt(4 * count)));
    return out;
  }
  downloadBytes(count: number, byteOffset = 0): Uint8Array {
    const out = new Uint8Array(count);
    if (count) check(hip.tstwo_download(ptr(out), this.dev + BigInt(byteOffset), BigInt(count)));
    return out;
  }
  /** tstwo_upload_async: the copy runs on the library's copy streams beside the kernels.  `words` should be page-locked (PinnedU32
   *  or hostRegister) and must stay alive and unchanged until uploadWait() / tstwo_sync. */
  uploadAsync(words: Uint32Array, byteOffset = 0): void {
    if (words.byteLength) check(hip.tstwo_upload_async(this.dev + BigInt(byteOffset), ptr(words), BigInt(words.byteLength)));
  }
  free(): void { check(hip.tstwo_free(this.dev)); }
}

/** Host hand-over beside the kernels (createBaseFieldColumn(data), backend/index.ts:20-31): page-lock a typed array the caller owns ... */
export function hostRegister(words: Uint32Array): void { ensureInit(); check(hip.tstwo_host_register(ptr(words), BigInt(words.byteLength))); }
export function hostUnregister(words: Uint32Array): void { check(hip.tstwo_host_unregister(ptr(words))); }
/** ... or take page-locked memory from the library: `view` is a Uint32Array over it. */
export class PinnedU32 {
  readonly view: Uint32Array;
  private addr: number;
  constructor(count: number) {
    ensureInit();
    const out = new BigUint64Array(1);
    check(hip.tstwo_host_alloc(ptr(out), BigInt(4 * count)));
    this.addr = Number(out[0]!);
    this.view = new Uint32Array(toArrayBuffer(this.addr as any, 0, 4 * count));
  }
  free(): void { check(hip.tstwo_host_free(this.addr as any)); }
}
/** Main-stream work enqueued after this call waits (on the device) for the copies issued so far; the host does not block. */
export function uploadFence(): void { check(hip.tstwo_upload_fence()); }
/** The host blocks until the copies issued so far have landed (their sources may be reused). */
export function uploadWait(): void { check(hip.tstwo_upload_wait()); }

/** Several small device buffers in ONE round trip (tstwo_download_many): pieces = [device address, words]. */
export function downloadMany(pieces: readonly (readonly [bigint, number])[]): Uint32Array[] {
  const srcs = new BigUint64Array(pieces.map(([dev]) => dev));
  const sizes = new BigUint64Array(pieces.map(([, words]) => BigInt(4 * words)));
  const total = pieces.reduce((acc, [, words]) => acc + words, 0);
  const out = new Uint32Array(total);
  if (total) check(hip.tstwo_download_many(ptr(srcs), ptr(sizes), BigInt(pieces.length), ptr(out)));
  const res: Uint32Array[] = [];
  let off = 0;
  for (const [, words] of pieces) { res.push(out.subarray(off, off + words)); off += words; }
  return res;
}

export const ptrs = (devs: bigint[]): BigUint64Array => BigUint64Array.from(devs.length ? devs : [0n]);
export const u32s = (vals: number[]): Uint32Array => Uint32Array.from(vals.length ? vals : [0]);
export { ptr };
