"""CPU oracle for the tensor-ops `runTOp`/`gradTOp` hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.
The product path (`tensor-ops_amd/`) never imports, links or calls anything in
this directory and fails loudly when its HIP extension is missing.

PARITY UNPINNED.  The reference (mstksg/tensor-ops, Haskell) ships no tests,
no golden vectors and no known-answer data (`test/Spec.hs:1-2` is a stub), and
it cannot be compiled here (no ghc/cabal/stack; un-vendored hmatrix / ad /
type-combinators).  There is therefore no `oracle/_ref` build.  The oracle is a
restatement of the reference *source*, each function citing the file:line it
follows, validated by
  * hand-derived exact known-answer tests on small integer tensors
    (`tests/test_oracle_nested.py`),
  * "BLAS dispatch == nested definition" identities
    (`src/TensorOps/Backend/BTensor.hs:141-175,592-716` vs
    `src/Data/Nested.hs:451-473`),
  * central finite differences of every `gradTOp'` against its `runTOp`,
  * an independent numpy `einsum`/`tensordot` formulation whose outputs are
    committed under `tests/golden/` together with the generating script.

Layout
  nested.py     Data.Nested semantics: gmul', transpose', index order, sumRows
  ad.py         forward-mode dual numbers = what `Numeric.AD.diff`/`grad` yield
  tensor.py     `class Tensor` (Types.hs:52-109) over numpy arrays (NTensor-like)
  top.py        `TOp`, Category, firstOp/secondOp/*>>/***/&&&, op vocabulary
  neuralnet.py  logistic/softmax/losses, Network, ffLayer, netGrad, trainNetwork
  btensor.py    `BTensor v b` (Backend/BTensor.hs) over ANY `class BLAS` dictionary: its rank dispatch (gmulB / gmulBLAS /
                dispatchBLAS / naiveGMul / liftBTensor / sumBTensor / transpBTensor), the HMat instance (BLAS/HMat.hs),
                and `instance Tensor (BTensor v b)` in the shape top.py takes a backend
  hmat_path.c   plain-C restatement of the BTensor->HMat BLAS call sequence for
                one ffLayer-stack gradTOp step (the single-core CPU baseline)
"""
