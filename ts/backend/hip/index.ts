// index.ts — HipBackend: the MI355X twin of CpuBackend behind tstwo's Backend / Column / PolyOps / FriOps / MerkleOps
// surface.  Intended location: packages/core/src/backend/hip/index.ts (imports below are relative to that place).
// NOT TESTED in the build image (no Bun); it is a line-for-line transcription of the Python mirror in tstwo_amd/
// (backend.py, poly.py, fri.py, vcs.py), which IS tested against the CPU oracle on an MI355X.
import { M31 } from "../../fields/m31";
import { QM31 } from "../../fields/qm31";
import type { Backend, Column } from "../index";
import type { MerkleOps } from "../../vcs/ops";
import { Blake2sHash } from "../../vcs/blake2_hash";
import { quotientConstants, type ColumnSampleBatch } from "../cpu/quotients";
import { CirclePoint, Coset } from "../../circle";
import type { CircleDomain } from "../../poly/circle/domain";
import { CirclePoly } from "../../poly/circle/poly";
import { CircleEvaluation, type BitReversedOrder } from "../../poly/circle/evaluation";
import { TwiddleTree } from "../../poly/twiddles";
import { LineDomain, LineEvaluation } from "../../poly/line";
import { SecureEvaluation } from "../../poly/circle/secure_poly";
import { bitReverseIndex } from "../../utils";
import { DeviceBuffer, check, ensureInit, hip, ptr, ptrs, u32s } from "./ffi";

const q4 = (q: QM31): Uint32Array => Uint32Array.from(q.to_m31_array().map((m) => m.value));

/** Column<M31> resident in HBM as little-endian u32 words (M31.intoSlice layout). */
export class HipColumn implements Column<M31> {
  constructor(readonly buf: DeviceBuffer, private readonly n: number) {}
  static fromArray(data: readonly M31[]): HipColumn {           // value semantics: copies (cpu/index.ts:89)
    const b = new DeviceBuffer(4 * data.length);
    b.upload(Uint32Array.from(data, (m) => m.value));
    return new HipColumn(b, data.length);
  }
  static zeros(len: number): HipColumn {
    const b = new DeviceBuffer(4 * len);
    if (len) check(hip.tstwo_zero(b.dev, BigInt(4 * len)));
    return new HipColumn(b, len);
  }
  static uninitialized(len: number): HipColumn { return new HipColumn(new DeviceBuffer(4 * len), len); }
  // Column<T> declares zeros / uninitialized as INSTANCE methods (backend/index.ts:54-58); CpuColumn throws in them, here they work
  zeros(len: number): HipColumn { return HipColumn.zeros(len); }
  uninitialized(len: number): HipColumn { return HipColumn.uninitialized(len); }
  /** Array-style length, so that code written against `readonly BaseField[]` columns (MerkleProver.commit sorts and filters
   *  by `c.length`, vcs/prover.ts:20-25) can be handed device columns. */
  get length(): number { return this.n; }
  get dev(): bigint { return this.buf.dev; }
  clone(): HipColumn {
    const out = HipColumn.uninitialized(this.n);
    if (this.n) check(hip.tstwo_copy(out.dev, this.dev, BigInt(4 * this.n)));
    return out;
  }
  len(): number { return this.n; }
  isEmpty(): boolean { return this.n === 0; }
  toU32(): Uint32Array { return this.buf.downloadU32(this.n); }
  toCpu(): M31[] { return Array.from(this.toU32(), (v) => M31.from_u32_unchecked(v)); }
  private checkIndex(i: number): void {
    if (!Number.isInteger(i) || i < 0 || i >= this.n) throw new Error(`Index ${i} out of bounds for column of length ${this.n}`);
  }
  at(i: number): M31 { this.checkIndex(i); return M31.from_u32_unchecked(this.buf.downloadU32(1, 4 * i)[0]!); }
  set(i: number, v: M31): void { this.checkIndex(i); this.buf.upload(Uint32Array.of(v.value), 4 * i); }
}

/** SecureColumnByCoords on the device: 4 coordinate columns (fields/secure_columns.ts:124); a Column<QM31> (backend/index.ts:53-74). */
export class HipSecureColumn implements Column<QM31> {
  constructor(readonly columns: [HipColumn, HipColumn, HipColumn, HipColumn]) {
    if (!columns.every((c) => c.len() === columns[0].len())) throw new Error("coordinate column length mismatch");
  }
  static from(values: readonly QM31[]): HipSecureColumn {
    const cols = [0, 1, 2, 3].map((k) => HipColumn.fromArray(values.map((q) => q.to_m31_array()[k]!)));
    return new HipSecureColumn(cols as [HipColumn, HipColumn, HipColumn, HipColumn]);
  }
  static zeros(n: number): HipSecureColumn { return new HipSecureColumn([0, 1, 2, 3].map(() => HipColumn.zeros(n)) as [HipColumn, HipColumn, HipColumn, HipColumn]); }
  static uninitialized(n: number): HipSecureColumn { return new HipSecureColumn([0, 1, 2, 3].map(() => HipColumn.uninitialized(n)) as [HipColumn, HipColumn, HipColumn, HipColumn]); }
  zeros(len: number): HipSecureColumn { return HipSecureColumn.zeros(len); }
  uninitialized(len: number): HipSecureColumn { return HipSecureColumn.uninitialized(len); }
  len(): number { return this.columns[0].len(); }
  isEmpty(): boolean { return this.len() === 0; }
  is_empty(): boolean { return this.isEmpty(); }                       // SecureColumnByCoords spelling (secure_columns.ts:154)
  ptrs(): BigUint64Array { return ptrs(this.columns.map((c) => c.dev)); }
  at(i: number): QM31 { return QM31.from_m31_array(this.columns.map((c) => c.at(i)) as [M31, M31, M31, M31]); }
  set(i: number, v: QM31): void { v.to_m31_array().forEach((m, k) => this.columns[k]!.set(i, m)); }
  to_vec(): QM31[] {
    const c = this.columns.map((x) => x.toU32());
    return Array.from({ length: this.len() }, (_, i) => QM31.from_u32_unchecked(c[0]![i]!, c[1]![i]!, c[2]![i]!, c[3]![i]!));
  }
  toCpu(): QM31[] { return this.to_vec(); }
}

export class HipBackend implements Backend {
  readonly name = "HipBackend";
  constructor() { ensureInit(); }
  bitReverseColumn(col: HipColumn): void {                           // throws "length is not power of two"
    check(hip.tstwo_bit_reverse(ptr(ptrs([col.dev])), 1n, BigInt(col.len())));
  }
  createBaseFieldColumn(data: M31[]): HipColumn { return HipColumn.fromArray(data); }
  createSecureFieldColumn(data: QM31[]): HipSecureColumn { return HipSecureColumn.from(data); }
  batchInverse(col: HipColumn): HipColumn {                          // throws "0 has no inverse"
    const out = HipColumn.uninitialized(col.len());
    check(hip.tstwo_m31_batch_inverse(col.dev, out.dev, BigInt(col.len())));
    return out;
  }
  batchInverseSecure(col: HipSecureColumn): HipSecureColumn {
    const out = HipSecureColumn.uninitialized(col.len());
    check(hip.tstwo_qm31_batch_inverse(ptr(col.ptrs()), ptr(out.ptrs()), BigInt(col.len())));
    return out;
  }
  /** Many small inversions in one phase: enqueue them with the *_async entry points and ask ONCE (tstwo_check_zero_flag). */
  batchInverseDeferred(col: HipColumn): HipColumn {
    const out = HipColumn.uninitialized(col.len());
    check(hip.tstwo_m31_batch_inverse_async(col.dev, out.dev, BigInt(col.len())));
    return out;
  }
  checkNoZeroInverse(): void { check(hip.tstwo_check_zero_flag()); }     // throws "0 has no inverse"
  // QuotientOps / AccumulationOps (backend/index.ts:82-91 leaves them empty; the functions live in backend/cpu/*.ts)
  accumulateQuotients = accumulateQuotients;
  accumulate = accumulate;
  generate_secure_powers = generate_secure_powers;
}

/** accumulateQuotients (backend/cpu/quotients.ts:52-75) on device columns.  The per-batch constants come from the reference's OWN
 *  quotientConstants() and the denominators are read the way its denominatorInverses() reads them (Pr / Pi = c0.real / c0.imag,
 *  quotients.ts:160-178), so the rows equal CpuBackend's bit for bit; `rustSemantics` switches to stwo's definitions
 *  (conj(a + bu) = a - bu, Pr = c0, Pi = c1), computed inside the library (tstwo_quotients_accumulate_samples). */
export function accumulateQuotients(
  domain: CircleDomain,
  columns: Array<HipCircleEvaluation>,
  random_coeff: QM31,
  sample_batches: ColumnSampleBatch[],
  _log_blowup_factor: number,
  rustSemantics = false,
): SecureEvaluation<HipBackend, BitReversedOrder> {
  const n = domain.size();
  columns.forEach((c) => { if (c.dev.len() !== n) throw new Error("column length does not match the domain size"); });
  const out = HipSecureColumn.uninitialized(n);
  const off = [0], cidx: number[] = [];
  sample_batches.forEach((sb) => { sb.columns_and_values.forEach(([ci]) => cidx.push(ci)); off.push(cidx.length); });
  const colPtrs = ptr(ptrs(columns.map((c) => c.dev.dev)));
  if (rustSemantics) {
    const points = sample_batches.flatMap((sb) => [...q4(sb.point.x), ...q4(sb.point.y)]);
    const values = sample_batches.flatMap((sb) => sb.columns_and_values.flatMap(([, v]) => [...q4(v)]));
    check(hip.tstwo_quotients_accumulate_samples(domain.halfCoset.initial_index.value, domain.log_size(), colPtrs, BigInt(columns.length),
      BigInt(sample_batches.length), ptr(u32s(off)), ptr(u32s(cidx)), ptr(u32s(points)), ptr(u32s(values)), ptr(q4(random_coeff)), ptr(out.ptrs())));
    return new SecureEvaluation(domain, out as any);
  }
  const qc = quotientConstants(sample_batches, random_coeff);
  const abc = qc.line_coeffs.flatMap((lc) => lc.flatMap(([a, b, c]) => [...q4(a), ...q4(b), ...q4(c)]));
  const bco = qc.batch_random_coeffs.flatMap((c) => [...q4(c)]);
  const real = (m: M31): number[] => [m.value, 0];                      // an M31 as a CM31 (value, 0)
  const prx = sample_batches.flatMap((sb) => real(sb.point.x.c0.real)), pry = sample_batches.flatMap((sb) => real(sb.point.y.c0.real));
  const pix = sample_batches.flatMap((sb) => real(sb.point.x.c0.imag)), piy = sample_batches.flatMap((sb) => real(sb.point.y.c0.imag));
  check(hip.tstwo_quotients_accumulate(domain.halfCoset.initial_index.value, domain.log_size(), colPtrs, BigInt(columns.length),
    BigInt(sample_batches.length), ptr(u32s(off)), ptr(u32s(cidx)), ptr(u32s(abc)), ptr(u32s(bco)),
    ptr(u32s(prx)), ptr(u32s(pry)), ptr(u32s(pix)), ptr(u32s(piy)), ptr(out.ptrs())));        // throws "0 has no inverse"
  return new SecureEvaluation(domain, out as any);
}

/** AccumulationOps.accumulate (backend/cpu/accumulation.ts:38-49): column[i] += other[i]. */
export function accumulate(column: HipSecureColumn, other: HipSecureColumn): void {
  if (column.len() !== other.len()) throw new Error("column length mismatch");
  check(hip.tstwo_secure_accumulate(ptr(column.ptrs()), ptr(other.ptrs()), BigInt(column.len())));
}

/** generate_secure_powers (accumulation.ts:52-63): a handful of scalars, host side like the reference. */
export function generate_secure_powers(felt: QM31, nPowers: number): QM31[] {
  const res: QM31[] = [];
  let acc = QM31.one();
  for (let i = 0; i < nPowers; i++) { res.push(acc); acc = acc.mul(felt); }
  return res;
}

/** TwiddleTree with device buffers; generated on the GPU (backend/cpu/circle.ts:210-239). */
export function precomputeTwiddles(coset: Coset): TwiddleTree<HipBackend, HipColumn> {
  const n = coset.size();
  const tw = HipColumn.uninitialized(n), itw = HipColumn.uninitialized(n);
  check(hip.tstwo_twiddles_build(coset.initial_index.value, coset.log_size, tw.dev, itw.dev));
  return new TwiddleTree(coset, tw, itw);
}

function checkTree(domain: CircleDomain, tw: TwiddleTree<HipBackend, HipColumn>): void {
  if (!domain.halfCoset.is_doubling_of(tw.rootCoset)) throw new Error("twiddle tree mismatch");
}

// @ts-expect-error static-method dispatch contract of the reference (test/poly/circleEvaluation.test.ts:5-16)
export class HipCircleEvaluation extends CircleEvaluation<HipBackend, M31, BitReversedOrder> {
  constructor(domain: CircleDomain, readonly dev: HipColumn) { super(domain, dev as any); }
  static precomputeTwiddles = precomputeTwiddles;
  static to_cpu(values: HipColumn): M31[] { return values.toCpu(); }
  static bitReverseColumn(col: HipColumn): void { new HipBackend().bitReverseColumn(col); }
}

export class HipCirclePoly extends CirclePoly<HipBackend> {
  /** true reproduces the reference's log_size == 3 output swap (backend/cpu/circle.ts:123-131); default = Rust-exact. */
  static compatLog3Swap = false;
  constructor(readonly dev: HipColumn) { super(dev as any); }
  static precomputeTwiddles = precomputeTwiddles;

  static extend(poly: HipCirclePoly, logSize: number): HipCirclePoly {
    if (logSize < poly.logSize()) throw new Error("log size too small");
    const out = HipColumn.uninitialized(1 << logSize);
    check(hip.tstwo_poly_extend(poly.dev.dev, poly.logSize(), out.dev, logSize));
    return new HipCirclePoly(out);
  }
  static evaluate(poly: HipCirclePoly, domain: CircleDomain, tw: TwiddleTree<HipBackend, HipColumn>): HipCircleEvaluation {
    return HipCirclePoly.evaluatePolynomials([poly], domain, tw)[0]!;
  }
  /** Value semantics (the evaluation survives) with the copy folded into the first pass (tstwo_cfft_interpolate_to). */
  static interpolate(ev: HipCircleEvaluation, tw: TwiddleTree<HipBackend, HipColumn>): HipCirclePoly {
    checkTree(ev.domain, tw);
    const n = ev.domain.log_size();
    const col = HipColumn.uninitialized(1 << n);
    check(hip.tstwo_cfft_interpolate_to(ptr(ptrs([ev.dev.dev])), ptr(ptrs([col.dev])), 1n, n, ev.domain.halfCoset.initial_index.value, tw.itwiddles.dev, tw.rootCoset.log_size));
    return new HipCirclePoly(col);
  }
  /** PolyOps.evaluatePolynomials, batched (poly/circle/ops.ts:89-101): extend + evaluate per group of equal-sized polynomials
   *  without materialising the zero padding (tstwo_cfft_evaluate_extended). */
  static evaluatePolynomials(polys: HipCirclePoly[], domain: CircleDomain, tw: TwiddleTree<HipBackend, HipColumn>): HipCircleEvaluation[] {
    checkTree(domain, tw);
    const n = domain.log_size();
    const outs = polys.map((p) => {
      if (n < p.logSize()) throw new Error("log size too small");
      return HipColumn.uninitialized(1 << n);
    });
    const byLog = new Map<number, number[]>();
    polys.forEach((p, i) => byLog.set(p.logSize(), [...(byLog.get(p.logSize()) ?? []), i]));
    for (const [lg, idxs] of byLog) {
      check(hip.tstwo_cfft_evaluate_extended(ptr(ptrs(idxs.map((i) => polys[i]!.dev.dev))), lg, ptr(ptrs(idxs.map((i) => outs[i]!.dev))),
        BigInt(idxs.length), n, domain.halfCoset.initial_index.value, tw.twiddles.dev, tw.rootCoset.log_size));
    }
    return outs.map((c) => new HipCircleEvaluation(domain, c));
  }
  /** All polynomials of one size at one point in one launch sequence (prove_values' out-of-domain sampling). */
  static evalAtPointBatch(polys: HipCirclePoly[], point: CirclePoint<QM31>): QM31[] {
    const out = new Uint32Array(4 * polys.length);
    check(hip.tstwo_eval_at_point_batch(ptr(ptrs(polys.map((p) => p.dev.dev))), BigInt(polys.length), polys[0]!.logSize(), ptr(q4(point.x)), ptr(q4(point.y)), ptr(out)));
    return polys.map((_, i) => QM31.from_u32_unchecked(out[4 * i]!, out[4 * i + 1]!, out[4 * i + 2]!, out[4 * i + 3]!));
  }
  static eval_at_point(poly: HipCirclePoly, point: CirclePoint<QM31>): QM31 {
    const out = new Uint32Array(4);
    check(hip.tstwo_eval_at_point(poly.dev.dev, poly.logSize(), ptr(q4(point.x)), ptr(q4(point.y)), ptr(out)));
    return QM31.from_u32_unchecked(out[0]!, out[1]!, out[2]!, out[3]!);
  }
}

/** FriOps (fri.ts:93-110) with the reference's error texts. */
export class HipFriOps {
  fold_line(ev: { values: HipSecureColumn; domain(): LineDomain; len(): number }, alpha: QM31, tw?: TwiddleTree<HipBackend, HipColumn>) {
    const n = ev.len();
    if (n < 2) throw new Error("fold_line: Evaluation too small, must have at least 2 elements.");
    const domain = ev.domain(), k = domain.logSize();
    const out = HipSecureColumn.uninitialized(n / 2);
    if (tw && domain.coset().is_doubling_of(tw.rootCoset)) {
      check(hip.tstwo_fri_fold_line(ptr(ev.values.ptrs()), k, tw.itwiddles.dev, tw.rootCoset.log_size, ptr(q4(alpha)), ptr(out.ptrs())));
    } else {   // domain not covered by a precomputed tree: the n/2 inverses the reference computes per element
      const inv = HipColumn.fromArray(Array.from({ length: n / 2 }, (_, i) => domain.at(bitReverseIndex(i << 1, k)).inverse()));
      check(hip.tstwo_fri_fold_line_tw(ptr(ev.values.ptrs()), k, inv.dev, ptr(q4(alpha)), ptr(out.ptrs())));
    }
    return LineEvaluation.new(domain.double(), out as any);
  }
  fold_circle_into_line(dst: { values: HipSecureColumn; len(): number }, src: { values: HipSecureColumn; domain: CircleDomain }, alpha: QM31, tw?: TwiddleTree<HipBackend, HipColumn>): void {
    if ((src.domain.size() >> 1) !== dst.len()) throw new Error("fold_circle_into_line: Length mismatch between src and dst after considering fold step.");
    const n = src.domain.log_size();
    if (tw && n >= 3 && src.domain.halfCoset.is_doubling_of(tw.rootCoset)) {
      check(hip.tstwo_fri_fold_circle_into_line(ptr(dst.values.ptrs()), BigInt(dst.len()), ptr(src.values.ptrs()), n, tw.itwiddles.dev, tw.rootCoset.log_size, ptr(q4(alpha))));
    } else {
      const inv = HipColumn.fromArray(Array.from({ length: dst.len() }, (_, i) => src.domain.at(bitReverseIndex(i << 1, n)).y.inverse()));
      check(hip.tstwo_fri_fold_circle_into_line_tw(ptr(dst.values.ptrs()), BigInt(dst.len()), ptr(src.values.ptrs()), n, inv.dev, ptr(q4(alpha))));
    }
  }
  /** LineEvaluation.interpolate (poly/line.ts:312-329) on the device for a layer of at most 2^12 values on a doubling of the
   *  tree's root — the last FRI layer: returns the four coefficient columns (bit-reversed coefficient order) still in HBM; fetch
   *  them together with the channel state in one `downloadMany` (ffi.ts). */
  line_interpolate(ev: { values: HipSecureColumn; domain(): LineDomain; len(): number }, tw: TwiddleTree<HipBackend, HipColumn>): HipSecureColumn {
    const domain = ev.domain(), k = domain.logSize();
    if (k > 12) throw new Error("line_interpolate: at most 2^12 values on the device");
    if (k >= 1 && !domain.coset().is_doubling_of(tw.rootCoset)) throw new Error("Not enough twiddles!");
    const out = HipSecureColumn.uninitialized(ev.len());
    check(hip.tstwo_line_interpolate(ptr(ev.values.ptrs()), k, tw.itwiddles.dev, tw.rootCoset.log_size, ptr(out.ptrs())));
    return out;
  }
  decompose(ev: { values: HipSecureColumn; domain: CircleDomain }): [SecureEvaluation<HipBackend, BitReversedOrder>, QM31] {
    const n = ev.values.len();
    const out = HipSecureColumn.uninitialized(n), lam = new Uint32Array(4);
    check(hip.tstwo_fri_decompose(ptr(ev.values.ptrs()), BigInt(n), ptr(out.ptrs()), ptr(lam)));
    return [new SecureEvaluation(ev.domain, out as any), QM31.from_u32_unchecked(lam[0]!, lam[1]!, lam[2]!, lam[3]!)];
  }
}

/** MerkleOps<Blake2sHash> EXACTLY as vcs/ops.ts:16-26 declares it, so `MerkleProver.commit(new HipMerkleOps(), columns)`
 *  (vcs/prover.ts:13-30) type-checks and runs unchanged: hashes and column values cross the boundary as host objects, one layer
 *  per call (hashNode semantics: children AND the layer's column values, vcs/blake2_merkle.ts:9-24).  A column that is a
 *  HipColumn (see its `length` getter) is hashed where it lives; a plain M31[] is uploaded first.  This is the compatibility
 *  path — a prover that keeps its trace in HBM uses HipMerkleProver below, which never brings a layer to the host. */
export class HipMerkleOps implements MerkleOps<Blake2sHash> {
  commitOnLayer(logSize: number, prevLayer: readonly Blake2sHash[] | undefined, columns: readonly (readonly M31[])[]): Blake2sHash[] {
    const n = 1 << logSize;
    const devCols = columns.map((c) => ((c as unknown) instanceof HipColumn ? (c as unknown as HipColumn) : HipColumn.fromArray(c)));
    devCols.forEach((c) => { if (c.len() !== n) throw new Error("column length does not match the layer size"); });
    let prev: DeviceBuffer | undefined;
    if (prevLayer !== undefined) {
      if (prevLayer.length !== 2 * n) throw new Error("previous layer must have twice as many hashes");
      const bytes = new Uint8Array(64 * n);
      prevLayer.forEach((h, i) => bytes.set(h.bytes, 32 * i));
      prev = new DeviceBuffer(64 * n);
      prev.upload(bytes);
    }
    const out = this.commitOnLayerDevice(logSize, prev, devCols);
    const flat = out.downloadBytes(32 * n);
    return Array.from({ length: n }, (_, i) => new Blake2sHash(flat.subarray(32 * i, 32 * i + 32)));
  }
  /** The same on device buffers (prevLayer: 2^(logSize+1) digests of 32 bytes; result: 2^logSize digests). */
  commitOnLayerDevice(logSize: number, prevLayer: DeviceBuffer | undefined, columns: readonly HipColumn[]): DeviceBuffer {
    const out = new DeviceBuffer(32 << logSize);
    check(hip.tstwo_merkle_commit_layer(logSize, prevLayer ? prevLayer.dev : 0n, ptr(ptrs(columns.map((c) => c.dev))), BigInt(columns.length), out.dev));
    return out;
  }
}

/** Device-resident replacement for MerkleProver<Blake2sHash> (vcs/prover.ts): commit / root / decommit with every layer kept in
 *  HBM (layer k at byte 32*(2^k-1) of one buffer).  Columns of mixed sizes join at their layer, input order kept within a size
 *  class, like MerkleProver.commit's stable sort. */
export class HipMerkleProver {
  private constructor(readonly layers: DeviceBuffer, readonly maxLog: number, private readonly rootBytes: Uint8Array) {}
  /** A tree the library built and handed out (tstwo_fri_commit_layers): `layers` in tstwo_merkle_commit's layout, root first. */
  static adopt(layers: DeviceBuffer, maxLog: number): HipMerkleProver { return new HipMerkleProver(layers, maxLog, layers.downloadBytes(32)); }
  static commit(columns: readonly HipColumn[]): HipMerkleProver {
    const logs = columns.map((c) => {
      const lg = Math.log2(c.len());
      if (!Number.isInteger(lg)) throw new Error("length is not power of two");
      return lg;
    });
    const maxLog = columns.length ? Math.max(...logs) : 0;
    const layers = new DeviceBuffer(32 * ((2 << maxLog) - 1)), root = new Uint8Array(32);
    check(hip.tstwo_merkle_commit(ptr(ptrs(columns.map((c) => c.dev))), ptr(u32s(logs)), BigInt(columns.length), layers.dev, ptr(root)));
    return new HipMerkleProver(layers, maxLog, root);
  }
  /** Several trees in one launch sequence (tstwo_merkle_commit_many: a TreeVec committed together, pcs/prover.ts:62-64): the same
   *  trees as commit() one by one; equally shaped trees (16 / 32 / 48 / 64 columns of one size) share their launches. */
  static commitMany(columnSets: readonly (readonly HipColumn[])[]): HipMerkleProver[] {
    const n = columnSets.length;
    const reqs = new BigUint64Array(4 * n), roots = new Uint8Array(32 * Math.max(n, 1));
    const keep: unknown[] = [], bufs: DeviceBuffer[] = [], maxLogs: number[] = [];
    columnSets.forEach((cols, r) => {
      const logs = cols.map((c) => Math.log2(c.len()));
      if (!logs.every(Number.isInteger)) throw new Error("length is not power of two");
      const maxLog = cols.length ? Math.max(...logs) : 0;
      const layers = new DeviceBuffer(32 * ((2 << maxLog) - 1)), cp = ptrs(cols.map((c) => c.dev)), lg = u32s(logs);
      keep.push(cp, lg); bufs.push(layers); maxLogs.push(maxLog);
      reqs[4 * r] = BigInt(ptr(cp)); reqs[4 * r + 1] = BigInt(ptr(lg)); reqs[4 * r + 2] = BigInt(cols.length); reqs[4 * r + 3] = layers.dev;
    });
    check(hip.tstwo_merkle_commit_many(ptr(reqs), BigInt(n), ptr(roots)));
    return bufs.map((b, r) => new HipMerkleProver(b, maxLogs[r]!, roots.slice(32 * r, 32 * r + 32)));
  }
  root(): Blake2sHash { return new Blake2sHash(this.rootBytes); }
  /** Device address of the root: tstwo_channel_mix_root_draw_felt / tstwo_allgather_roots read it without a host round trip. */
  rootDev(): bigint { return this.layers.dev; }
  /** MerkleProver.decommit (vcs/prover.ts:32-109) on a tree built by commit(): the walk and both gathers run in the library. */
  decommit(queriesPerLogSize: ReadonlyMap<number, number[]>, columns: readonly HipColumn[]):
      [M31[], { hashWitness: Blake2sHash[]; columnWitness: M31[] }] {
    const layers = this.layers;
    const logs = columns.map((c) => Math.log2(c.len()));
    const maxLog = this.maxLog;
    const sets = [...queriesPerLogSize].filter(([, q]) => q.length > 0);
    const totalQ = sets.reduce((a, [, q]) => a + q.length, 0);
    const capV = Math.max(1, totalQ * Math.max(1, columns.length)), capH = Math.max(1, 2 * totalQ * (maxLog + 1));
    const qArrays = sets.map(([, q]) => BigUint64Array.from(q.map(BigInt)));
    const qPtrs = BigUint64Array.from(qArrays.map((a) => BigInt(ptr(a))));
    const nQ = BigUint64Array.from(sets.map(([, q]) => BigInt(q.length)));
    const queried = new Uint32Array(capV), colWit = new Uint32Array(capV), hashes = new Uint8Array(32 * capH);
    const counts = BigUint64Array.from([BigInt(capV), BigInt(capH), BigInt(capV)]);
    check(hip.tstwo_merkle_decommit(layers.dev, maxLog, ptr(ptrs(columns.map((c) => c.dev))), ptr(u32s(logs)), BigInt(columns.length),
      ptr(u32s(sets.map(([lg]) => lg))), ptr(qPtrs), ptr(nQ), BigInt(sets.length),
      ptr(queried), ptr(counts.subarray(0, 1)), ptr(hashes), ptr(counts.subarray(1, 2)), ptr(colWit), ptr(counts.subarray(2, 3))));
    const [nq, nh, nw] = [Number(counts[0]), Number(counts[1]), Number(counts[2])];
    return [
      Array.from(queried.subarray(0, nq), (v) => M31.from_u32_unchecked(v)),
      { hashWitness: Array.from({ length: nh }, (_, i) => new Blake2sHash(hashes.slice(32 * i, 32 * i + 32))),
        columnWitness: Array.from(colWit.subarray(0, nw), (v) => M31.from_u32_unchecked(v)) },
    ];
  }
}

/** One committed FRI layer as tstwo_fri_decommit sees it: the tree and the evaluations under it (first layer: the circle
 *  evaluations, possibly of several sizes; inner layers: one line evaluation). */
export interface HipFriLayer { tree: HipMerkleProver; evaluations: readonly HipSecureColumn[]; }
export interface HipFriLayerProof { friWitness: QM31[]; hashWitness: Blake2sHash[]; columnWitness: M31[]; commitment: Blake2sHash; }
/** FriProver.decommit_on_queries (fri.ts:768-785) in ONE library call: per layer the position logic of
 *  computeDecommitmentPositionsAndWitnessEvals (fri.ts:346-384), the witness evaluations, the Merkle decommitment of the layer's
 *  tree and its root.  `queries`: ascending distinct positions in [0, 2^logDomainSize). */
export function friDecommit(layers: readonly HipFriLayer[], queries: readonly number[], logDomainSize: number,
                            firstFoldStep = 1, foldStep = 1): HipFriLayerProof[] {
  const n = layers.length, nq = queries.length;
  const desc = new BigUint64Array(5 * n);          // packed tstwo_fri_layer structs
  const keep: unknown[] = [];
  let totalEvals = 0, capH = 1;
  layers.forEach((l, r) => {
    const cols = ptrs(l.evaluations.flatMap((e) => e.columns.map((c) => c.dev)));
    const logs = u32s(l.evaluations.map((e) => Math.log2(e.len())));
    keep.push(cols, logs);
    desc[5 * r] = l.tree.rootDev(); desc[5 * r + 1] = BigInt(l.tree.maxLog); desc[5 * r + 2] = BigInt(ptr(cols));
    desc[5 * r + 3] = BigInt(ptr(logs)); desc[5 * r + 4] = BigInt(l.evaluations.length);
    totalEvals += l.evaluations.length;
    capH += 4 * nq * (l.tree.maxLog + 1);
  });
  let capE = Math.max(1, 2 * nq * totalEvals), capW = Math.max(1, 8 * nq * totalEvals);
  const roots = new Uint8Array(32 * n), counts = new BigUint64Array(3 * n);
  const qs = BigUint64Array.from(nq ? queries.map(BigInt) : [0n]);       // (ptr() of an empty typed array is not a valid address)
  let evals!: Uint32Array, hashes!: Uint8Array, colWit!: Uint32Array;
  // The capacities above are estimates: a first layer with several column sizes and sparse queries can exceed them.  The library
  // then fails with "output buffer too small" and leaves the EXACT totals behind — reallocate from them and call once more
  // (what tstwo_amd/fri_prover.py does).
  for (let attempt = 0; ; attempt++) {
    evals = new Uint32Array(4 * capE); hashes = new Uint8Array(32 * capH); colWit = new Uint32Array(capW);
    const totals = BigUint64Array.from([BigInt(capE), BigInt(capH), BigInt(capW)]);
    const rc = hip.tstwo_fri_decommit(ptr(desc), BigInt(n), ptr(qs), BigInt(nq), logDomainSize, firstFoldStep, foldStep,
      ptr(evals), ptr(hashes), ptr(colWit), ptr(roots), ptr(counts), ptr(totals));
    if (rc === 0) break;
    const msg = String(hip.tstwo_last_error());
    if (attempt > 0 || !msg.includes("output buffer too small")) throw new Error(msg);
    capE = Math.max(1, Number(totals[0])); capH = Math.max(1, Number(totals[1])); capW = Math.max(1, Number(totals[2]));
  }
  const out: HipFriLayerProof[] = [];
  let e0 = 0, h0 = 0, w0 = 0;
  for (let r = 0; r < n; r++) {
    const [ne, nh, nw] = [Number(counts[3 * r]), Number(counts[3 * r + 1]), Number(counts[3 * r + 2])];
    out.push({
      friWitness: Array.from({ length: ne }, (_, i) => QM31.from_u32_unchecked(evals[4 * (e0 + i)]!, evals[4 * (e0 + i) + 1]!, evals[4 * (e0 + i) + 2]!, evals[4 * (e0 + i) + 3]!)),
      hashWitness: Array.from({ length: nh }, (_, i) => new Blake2sHash(hashes.slice(32 * (h0 + i), 32 * (h0 + i) + 32))),
      columnWitness: Array.from(colWit.subarray(w0, w0 + nw), (v) => M31.from_u32_unchecked(v)),
      commitment: new Blake2sHash(roots.slice(32 * r, 32 * r + 32)),
    });
    e0 += ne; h0 += nh; w0 += nw;
  }
  return out;
}

/** One layer tstwo_fri_commit_layers produced: its line evaluation (4 coordinate columns of 2^logSize rows) and the tree over it
 *  (null for the last layer, which is interpolated, not committed). */
export interface HipFriCommittedLayer { logSize: number; columns: HipSecureColumn; tree: HipMerkleProver | null; }
/** FriProver.commit's layer loop (commitInnerLayers, fri.ts:676-716) in ONE library call: the first layer's tree over every
 *  circle evaluation's coordinate columns, then per layer mix_root / draw_felt on the device channel `chan` (10 words of device
 *  memory: upload the host channel's state before, download it after), fold, commit.  `circleEvals` in decreasing size;
 *  `alphas` receives the drawn challenges (16 bytes each).  Every buffer returned was allocated by the library (tstwo_malloc)
 *  and is owned by the returned objects. */
export function friCommitLayers(circleEvals: readonly HipSecureColumn[], itwiddles: HipColumn, twLog: number, logLastLayerSize: number,
                                chan: DeviceBuffer, alphas: DeviceBuffer): { firstTree: HipMerkleProver; layers: HipFriCommittedLayer[] } {
  const logs = circleEvals.map((e) => Math.log2(e.len()));
  const firstLog = logs[0]! - 1;
  if (firstLog < logLastLayerSize) throw new Error("last layer domain size mismatch");
  const cap = firstLog - logLastLayerSize + 1;
  const outs = new BigUint64Array(6 * cap);        // packed tstwo_fri_layer_out: { u32 log_size (+ pad); u32 *cols[4]; u8 *layers }
  const first = new BigUint64Array(1), nOut = new BigUint64Array(1);
  const cols = ptrs(circleEvals.flatMap((e) => e.columns.map((c) => c.dev))), lg = u32s(logs);
  check(hip.tstwo_fri_commit_layers(ptr(cols), ptr(lg), BigInt(circleEvals.length), itwiddles.dev, twLog, logLastLayerSize, chan.dev,
    alphas.dev, BigInt(Math.floor(alphas.nbytes / 16)), ptr(first), ptr(outs), BigInt(cap), ptr(nOut)));
  const layers: HipFriCommittedLayer[] = [];
  for (let i = 0; i < Number(nOut[0]); i++) {
    const logSize = Number(outs[6 * i]! & 0xffffffffn), n = 2 ** logSize;
    const columns = new HipSecureColumn([1, 2, 3, 4].map((k) => new HipColumn(DeviceBuffer.adopt(outs[6 * i + k]!, 4 * n), n)) as [HipColumn, HipColumn, HipColumn, HipColumn]);
    const treeDev = outs[6 * i + 5]!;
    layers.push({ logSize, columns, tree: treeDev ? HipMerkleProver.adopt(DeviceBuffer.adopt(treeDev, 32 * ((2 << logSize) - 1)), logSize) : null });
  }
  return { firstTree: HipMerkleProver.adopt(DeviceBuffer.adopt(first[0]!, 32 * ((2 << logs[0]!) - 1)), logs[0]!), layers };
}

/** The one exchange of a column-sharded prover (SURVEY.md 8e; include/tstwo_hip.h "multi-GPU"): one Bun process per GPU,
 *  the RCCL unique id handed from rank 0 to the others by any host channel (a file, a socket, an env variable). */
export class HipComm {
  private constructor(readonly rank: number, readonly world: number) {}
  static uniqueId(): Uint8Array { const id = new Uint8Array(128); check(hip.tstwo_comm_unique_id(ptr(id))); return id; }
  static init(rank: number, world: number, id: Uint8Array): HipComm {
    ensureInit();
    check(hip.tstwo_comm_init(rank, world, ptr(id)));
    return new HipComm(rank, world);
  }
  /** All-gather of the trees' roots in rank order = stwo's TreeVec order (pcs/prover.ts:62-64,227-228). */
  allgatherRoots(tree: HipMerkleProver): Blake2sHash[] {
    const out = new DeviceBuffer(32 * this.world);
    check(hip.tstwo_allgather_roots(tree.rootDev(), out.dev));
    const flat = out.downloadBytes(32 * this.world);
    out.free();
    return Array.from({ length: this.world }, (_, r) => new Blake2sHash(flat.subarray(32 * r, 32 * r + 32)));
  }
  close(): void { check(hip.tstwo_comm_destroy()); }
}
