{-# LANGUAGE DataKinds            #-}
{-# LANGUAGE GADTs                #-}
{-# LANGUAGE InstanceSigs         #-}
{-# LANGUAGE KindSignatures       #-}
{-# LANGUAGE LambdaCase           #-}
{-# LANGUAGE RankNTypes           #-}
{-# LANGUAGE ScopedTypeVariables  #-}
{-# LANGUAGE TypeApplications     #-}
{-# LANGUAGE TypeFamilies         #-}

-- | @instance BLAS HipB@: the MI355X backend behind the reference's INNER boundary, the one its README
-- prescribes for new backends ("make your type an instance of the BLAS typeclass ... and you get it for free",
-- @README.md:150-154@).  Replaces @src/TensorOps/BLAS/HMat.hs:103-231@ method for method; with it
-- @BTensor v HipB@ is a 'Tensor' through @src/TensorOps/Backend/BTensor.hs:775-879@ unchanged.  (The direct
-- @instance Tensor HipT@ of "TensorOps.Backend.HipTensor" is the faster integration: flat rank-N storage instead
-- of @BTensor@'s boxed nesting, and no GEMM-with-@eye@ matrix addition, @BTensor.hs:113@.)
--
-- NOT type-checked in this repository's build image (no GHC); see "TensorOps.HIP.FFI".
module TensorOps.BLAS.HIP
  ( HipB(..)
  ) where

import           Control.DeepSeq
import           Data.Kind
import           Data.Singletons
import           Data.Singletons.TypeLits
import           Data.Type.Combinator
import           Data.Type.Vector            (Vec, VecT(..))
import           Foreign
import           Foreign.C.Types
import           System.IO.Unsafe            (unsafePerformIO)
import           TensorOps.BLAS
import           TensorOps.HIP.Expr
import           TensorOps.HIP.FFI
import           Type.Class.Higher
import           Unsafe.Coerce                (unsafeCoerce)
import           Type.Class.Higher.Util
import qualified Data.Finite                 as DF
import qualified Data.Finite.Internal        as DF
import qualified Data.Vector.Storable        as VS

-- | A rank-1 or rank-2 device value; the shape index is a phantom (the handle carries the run-time dims).
newtype HipB (s :: BShape Nat) = HipB { unB :: H }

instance NFData (HipB s) where
    rnf (HipB h) = unsafePerformIO (forceH h)      -- enqueue, do not wait: later reads are stream-ordered
instance NFData1 HipB

instance Show (HipB s) where
    showsPrec p (HipB h) = showParen (p > 10) $ showString "HipB " . shows (shapeOf h)
instance Show1 HipB

toD :: E -> CDouble
toD (C a) = realToFrac a
toD (S _) = error "tensorops_hip: symbolic scalar where a number is needed (a closure captured a lifted variable)"

vecList :: Vec n a -> [a]
vecList = \case
    ØV        -> []
    I x :* xs -> x : vecList xs

dimsOfShape :: Sing (s :: BShape Nat) -> [Integer]
dimsOfShape = \case
    SBV n   -> [fromSing n]
    SBM n m -> [fromSing n, fromSing m]

-- | The symbolic variables a closure over @Vec n E@ is applied to.
symVec :: Vec n a -> Vec n E
symVec = go 0
  where
    go :: Int -> Vec m a -> Vec m E
    go _ ØV          = ØV
    go i (_ :* rest) = I (S (XVar i)) :* go (i + 1) rest

-- | Host traversal helpers for the Applicative-effectful methods (@BLAS.hs:140-159@): the effects are arbitrary
-- @f@, so they cannot run on the device -- one download, the same list traversal as @HMat.hs:177-218@, one upload.
rowsOf :: Int -> VS.Vector Double -> [VS.Vector Double]
rowsOf m v = [ VS.slice (i * m) m v | i <- [0 .. VS.length v `div` max 1 m - 1] ]

instance BLAS HipB where
    type ElemB HipB = E

    liftB
        :: forall n s. ()
        => Sing s
        -> (Vec n E -> E)
        -> Vec n (HipB s)
        -> HipB s
    liftB s f xs = case xs of
        ØV -> HipB $ unsafePerformIO $ do                           -- a constant of shape s (HMat.hs:115-119)
                dt <- elemDType
                withDims (dimsOfShape s) $ \r pd -> new1 (c_fill dt r pd 0 (toD (f ØV)))
        _  -> case f (symVec xs) of
                C c -> HipB $ unsafePerformIO $ do                  -- the closure ignores its arguments
                         dt <- elemDType
                         withDims (dimsOfShape s) $ \r pd -> new1 (c_fill dt r pd 0 (realToFrac c))
                S x -> let hx = compileX (length hs) x
                           hs = map unB (vecList xs)
                       in HipB $ unsafePerformIO $
                            withForeignPtr hx $ \pe -> withHs hs $ \k ph -> new1 (c_lift pe k ph)

    axpy a (HipB x) my = HipB $ unsafePerformIO $ withForeignPtr x $ \px ->
        case my of
          Nothing       -> new1 (c_axpy (toD a) px nullPtr)
          Just (HipB y) -> withForeignPtr y $ \py -> new1 (c_axpy (toD a) px py)

    dot (HipB x) (HipB y) = C . realToFrac $ unsafePerformIO $
        with2 x y $ \px py -> alloca $ \o -> chk (c_dot px py o) >> peek o

    ger (HipB x) (HipB y) = HipB $ unsafePerformIO $ with2 x y $ \px py -> new1 (c_ger px py)

    gemv a (HipB m) (HipB x) mby = HipB $ unsafePerformIO $ with2 m x $ \pm px ->
        case mby of
          Nothing            -> new1 (c_gemv (toD a) pm px 0 nullPtr)
          Just (b, HipB y)   -> withForeignPtr y $ \py -> new1 (c_gemv (toD a) pm px (toD b) py)

    gemm a (HipB m) (HipB n) mbc = HipB $ unsafePerformIO $ with2 m n $ \pm pn ->
        case mbc of
          Nothing            -> new1 (c_gemm (toD a) pm pn 0 nullPtr)
          Just (b, HipB c)   -> withForeignPtr c $ \pc -> new1 (c_gemm (toD a) pm pn (toD b) pc)

    scaleB a (HipB x) = HipB $ unsafePerformIO $ withForeignPtr x $ \px -> new1 (c_bscale (toD a) px)
    addB (HipB x) (HipB y) = HipB $ unsafePerformIO $ with2 x y $ \px py -> new1 (c_badd px py)

    indexB = \case
        PBV i   -> \(HipB x) -> idx x [DF.getFinite i]
        PBM i j -> \(HipB x) -> idx x [DF.getFinite i, DF.getFinite j]
      where
        idx x is = C . realToFrac $ unsafePerformIO $ withForeignPtr x $ \px ->
            withArray (map fromIntegral is) $ \pi' -> alloca $ \o -> chk (c_index px pi' 0 o) >> peek o

    indexRowB i (HipB m) = HipB $ unsafePerformIO $ withForeignPtr m $ \pm ->
        new1 (c_index_row (fromIntegral (DF.getFinite i)) pm)          -- a zero-copy view

    transpB (HipB m) = HipB $ unsafePerformIO $ withForeignPtr m $ \pm -> new1 (c_btransp pm)   -- a view, like `tr`

    iRowsB f (HipB m) =
        let ([n, c], _) = shapeOf m
            host        = unsafePerformIO (toHost m)
            rows        = rowsOf (fromIntegral c) host
            up k v      = HipB (unsafePerformIO (fromHost [k] 0 v))
            down (HipB r) = unsafePerformIO (toHost r)
            finish rs   = let vs = map down rs
                              o  = if null vs then 0 else VS.length (head vs)
                          in HipB (unsafePerformIO (fromHost [fromIntegral n, fromIntegral o] 0 (VS.concat vs)))
        in finish <$> traverse (\(i, r) -> f (DF.Finite i) (up (fromIntegral c) r)) (zip [0 ..] rows)

    iElemsB f (HipB x) =
        let (ds, _) = shapeOf x
            host    = VS.toList (unsafePerformIO (toHost x))
            back vs = HipB (unsafePerformIO (fromHost (map fromIntegral ds) 0 (VS.fromList (map (realToFrac . toD) vs))))
        in case ds of
             [_]    -> back <$> traverse (\(i, e) -> f (pbvu i) (C e)) (zip [0 ..] host)
             [_, c] -> back <$> traverse (\(k, e) -> f (pbmu (k `div` fromIntegral c) (k `mod` fromIntegral c)) (C e))
                                         (zip [0 ..] host)
             _      -> error "tensorops_hip: iElemsB on a handle that is neither a vector nor a matrix"
      where
        -- (the phantom shape of the argument fixes the constructor; the index values are run-time)
        pbvu i   = unsafeShapeIx [i]
        pbmu i j = unsafeShapeIx [i, j]

    bgenA s f = case s of
        SBV sN    -> (\es -> up [fromSing sN] es) <$> traverse (\i -> f (PBV (DF.Finite i))) [0 .. fromSing sN - 1]
        SBM sN sM -> (\es -> up [fromSing sN, fromSing sM] (concat es))
                       <$> traverse (\i -> traverse (\j -> f (PBM (DF.Finite i) (DF.Finite j))) [0 .. fromSing sM - 1])
                                    [0 .. fromSing sN - 1]
      where
        up ds es = HipB (unsafePerformIO (fromHost ds 0 (VS.fromList (map (realToFrac . toD) es))))

    bgenRowsA
        :: forall f n m. (Applicative f, SingI n)
        => (DF.Finite n -> f (HipB ('BV m)))
        -> f (HipB ('BM n m))
    bgenRowsA f = stackRows <$> traverse (f . DF.Finite) [0 .. n - 1]
      where
        n = fromSing (sing @Nat @n)
        stackRows rs = HipB $ unsafePerformIO $
            withDims [n] $ \r pd -> withHs (map unB rs) $ \_ ph -> new1 (c_stack r pd ph)

    eye sN = HipB $ unsafePerformIO $ do
        dt <- elemDType
        new1 (c_eye dt (fromIntegral (fromSing sN)))
    traceB (HipB m) = C . realToFrac $ unsafePerformIO $ withForeignPtr m $ \pm ->
        alloca $ \o -> chk (c_trace pm o) >> peek o
    diagB (HipB x)    = HipB $ unsafePerformIO $ withForeignPtr x $ \px -> new1 (c_bdiag px)
    getDiagB (HipB m) = HipB $ unsafePerformIO $ withForeignPtr m $ \pm -> new1 (c_bget_diag pm)
    sumB (HipB x) = C . realToFrac $ unsafePerformIO $ withForeignPtr x $ \px ->
        alloca $ \o -> chk (c_bsum px o) >> peek o

-- | Index witnesses built from run-time integers.  @Finite n@ is a newtype over 'Integer'
-- ("Data.Finite.Internal", used the same way by @HMat.hs:178,185,189@); the shape index is fixed by the caller's type.
unsafeShapeIx :: [Integer] -> BShapeP DF.Finite s
unsafeShapeIx = \case
    [i]    -> unsafeCoerceShape (PBV (DF.Finite i))
    [i, j] -> unsafeCoerceShape (PBM (DF.Finite i) (DF.Finite j))
    _      -> error "unsafeShapeIx"
  where
    unsafeCoerceShape :: BShapeP DF.Finite a -> BShapeP DF.Finite b
    unsafeCoerceShape = unsafeCoerce
